"""`Parameter`: what a backbone "keras_layer" receives as its configuration.

The plugin contract is `Layer(params, name, reuse=None)` (reference easy_rec/python/layers/utils.py:165-260 defines
the accessor set a layer may use: attribute / item access, `get_or_default`, `check_required`, `has_field`,
`make_from_pb`, `get_pb_config`, `l2_regularizer`).  A layer's settings arrive either as a typed proto message
(`keras_layer { mlp { ... } }`) or as a free-form `google.protobuf.Struct` (`st_params`); this class hides which.

Implementation: two small source adapters (`_StructSource`, `_MessageSource`) answer "is the key present", "fetch
it" and "is it a nested config"; `Parameter` itself only wraps nested configs and carries the L2 coefficient.
"""
from google.protobuf import struct_pb2
from google.protobuf.descriptor import FieldDescriptor

_MISSING = object()


def is_proto_message(pb_obj, field):
  """True when `field` of the message `pb_obj` is itself a message (a nested configuration)."""
  desc = getattr(pb_obj, 'DESCRIPTOR', None)
  fd = None if desc is None else desc.fields_by_name.get(field)
  return fd is not None and fd.type == FieldDescriptor.TYPE_MESSAGE


class _StructSource(object):
  """Settings held in a google.protobuf.Struct (numbers arrive as floats)."""

  def __init__(self, struct):
    self.raw = struct

  def fetch(self, key):
    return self.raw[key] if key in self.raw else _MISSING

  def present(self, key):
    return key in self.raw

  def nested(self, key, value):
    return isinstance(value, struct_pb2.Struct)

  def coerce(self, value, like):
    # Struct has only doubles: an integer default asks for an integer back
    if like is not None and isinstance(value, float):
      return type(like)(value)
    return value


class _MessageSource(object):
  """Settings held in a typed proto message."""

  def __init__(self, msg):
    self.raw = msg

  def fetch(self, key):
    return getattr(self.raw, key, _MISSING)

  def present(self, key):
    value = getattr(self.raw, key, _MISSING)
    if value is _MISSING:
      return False
    if hasattr(value, '__len__'):
      # repeated field: present when non-empty.  STRINGS have a length too, and the reference's test is exactly this
      # one (layers/utils.py:226-229): an unset string field whose proto default is non-empty counts as present and
      # yields that default - e.g. `mlp { hidden_units: ... }` gets final_activation 'relu' (protos/dnn.proto:26), where
      # the same MLP configured through st_params gets the layer's default None.  Pinned by
      # tests/test_reference_layers.py::test_product_keras_mmoe against the reference's own Parameter.
      return len(value) > 0
    try:
      return self.raw.HasField(key)
    except ValueError:  # proto3-style scalar without presence: treat as unset
      return False

  def nested(self, key, value):
    return is_proto_message(self.raw, key)

  def coerce(self, value, like):
    return value


class Parameter(object):

  def __init__(self, params, is_struct, l2_reg=None):
    self.__dict__['params'] = params
    self.__dict__['is_struct'] = bool(is_struct)
    self.__dict__['_l2_reg'] = l2_reg
    self.__dict__['_src'] = _StructSource(params) if is_struct else _MessageSource(params)

  # -- construction helpers
  @staticmethod
  def make_from_pb(config):
    return Parameter(config, False)

  def get_pb_config(self):
    if self.is_struct:
      raise AssertionError('Struct parameter can not convert to pb config')
    return self.params

  # -- the L2 coefficient rides along into nested configs
  @property
  def l2_regularizer(self):
    return self._l2_reg

  @l2_regularizer.setter
  def l2_regularizer(self, value):
    self.__dict__['_l2_reg'] = value

  def __setattr__(self, key, value):
    if key == 'l2_regularizer':
      self.__dict__['_l2_reg'] = value
    else:
      self.__dict__[key] = value

  # -- access
  def _wrap(self, key, value):
    if self._src.nested(key, value):
      return Parameter(value, self.is_struct, self._l2_reg)
    return value

  def __getattr__(self, key):
    if key.startswith('__'):  # copy / pickle probes are not configuration keys
      raise AttributeError(key)
    value = self._src.fetch(key)
    if value is _MISSING:
      if self.is_struct:
        return None
      raise AttributeError(key)
    return self._wrap(key, value)

  __getitem__ = __getattr__

  def get_or_default(self, key, def_val):
    """The configured value of `key`, or `def_val` when the config does not set it (an unset proto field does NOT
    fall back to the proto's own default: the layer's default wins - except for string fields, see
    _MessageSource.present)."""
    if not self._src.present(key):
      return def_val
    return self._src.coerce(self._src.fetch(key), def_val)

  def has_field(self, key):
    return key in self.params if self.is_struct else self.params.HasField(key)

  def check_required(self, keys):
    if not self.is_struct:
      return  # typed messages enforce their own required fields
    for key in ([keys] if isinstance(keys, str) else list(keys)):
      if not self._src.present(key):
        raise KeyError('%s must be set in params' % key)
