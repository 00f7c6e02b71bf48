"""easyrec_amd: an MI355X-native (gfx950) training path for EasyRec's sparse-embedding +
feature-interaction + MLP hot loop, behind EasyRec's protobuf-config API.

Layout mirrors the reference's package (easy_rec/python/...) for the parts of the hot
path it rebuilds: protos/ utils/ feature_column/ input/ layers/ model/ builders/ core/,
plus csrc/ (hand-written HIP kernels behind the C ABI declared in include/easyrec_hip.h).
"""
__version__ = '0.1.0'
