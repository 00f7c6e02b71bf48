"""Process-wide build context (the analogue of TF's default graph + variable scopes).

The reference relies on TF's implicit global state: `tf.get_variable`, the REGULARIZATION_LOSSES
and UPDATE_OPS collections (model/easy_rec_estimator.py:166-213).  Layers here fetch the dense
`VarStore` and the `EmbeddingEngine` of the model being built/executed from this context.
"""
import contextlib
import threading

_TLS = threading.local()  # one context stack per thread (tests run several ranks as threads)


def _stack():
  st = getattr(_TLS, 'stack', None)
  if st is None:
    st = _TLS.stack = []
  return st


class ModelContext(object):

  def __init__(self, varstore, engine, is_training=True):
    self.varstore = varstore
    self.engine = engine
    self.is_training = is_training
    self.building = True  # build pass: variables are being created, moving statistics frozen
    self.dense_dtype = 'f32'  # 'bf16': MLP / cross GEMMs on bf16 MFMA with fp32 accumulate (er_gemm_bf16)
    # per step (EasyRecModel.begin_step): logit heads waiting for the loss builder (kernels.HeadState, keyed by the logits'
    # data pointer) and the column-sum jobs their fused launch leaves to the loss tail [(partial [P, ld], dst, n_cols)]
    self.heads = {}
    self.tail_jobs = []
    # per step: one gradient buffer per activation tensor shared by its consumers' backward kernels (kernels.grad_slot)
    self.grad_slots = {}
    # dense_dtype 'bf16' on the GPU: the kernels.Bf16Shadows of the model (weight shadows + the step's bf16 operand copies)
    self.bf16_state = None


@contextlib.contextmanager
def use(ctx):
  _stack().append(ctx)
  try:
    yield ctx
  finally:
    _stack().pop()


def current():
  st = _stack()
  assert st, 'no active easyrec_amd ModelContext (wrap model calls in core.context.use(ctx))'
  return st[-1]


def varstore():
  return current().varstore
