"""`Parameter`: the uniform view over a layer's configuration - a typed proto message or a
google.protobuf.Struct - that every backbone "keras_layer" receives (the plugin contract
`Layer(params, name, reuse=None)`; reference easy_rec/python/layers/utils.py:165-260)."""
from google.protobuf import struct_pb2
from google.protobuf.descriptor import FieldDescriptor


def is_proto_message(pb_obj, field):
  if not hasattr(pb_obj, 'DESCRIPTOR'):
    return False
  if field not in pb_obj.DESCRIPTOR.fields_by_name:
    return False
  return pb_obj.DESCRIPTOR.fields_by_name[field].type == FieldDescriptor.TYPE_MESSAGE


class Parameter(object):

  def __init__(self, params, is_struct, l2_reg=None):
    self.params = params
    self.is_struct = is_struct
    self._l2_reg = l2_reg

  @staticmethod
  def make_from_pb(config):
    return Parameter(config, False)

  def get_pb_config(self):
    assert not self.is_struct, 'Struct parameter can not convert to pb config'
    return self.params

  @property
  def l2_regularizer(self):
    return self._l2_reg

  @l2_regularizer.setter
  def l2_regularizer(self, value):
    self._l2_reg = value

  def __getattr__(self, key):
    if key in ('params', 'is_struct', '_l2_reg'):
      raise AttributeError(key)
    if self.is_struct:
      if key not in self.params:
        return None
      value = self.params[key]
      if isinstance(value, struct_pb2.Struct):
        return Parameter(value, True, self._l2_reg)
      return value
    value = getattr(self.params, key)
    if is_proto_message(self.params, key):
      return Parameter(value, False, self._l2_reg)
    return value

  def __getitem__(self, key):
    return self.__getattr__(key)

  def get_or_default(self, key, def_val):
    if self.is_struct:
      if key in self.params:
        if def_val is None:
          return self.params[key]
        value = self.params[key]
        if isinstance(value, float):
          return type(def_val)(value)
        return value
      return def_val
    value = getattr(self.params, key, def_val)
    if hasattr(value, '__len__') and not isinstance(value, (str, bytes)):  # repeated
      return value if len(value) > 0 else def_val
    try:
      if self.params.HasField(key):
        return value
    except ValueError:
      pass
    return def_val  # maybe not equal to the default value of the msg field

  def check_required(self, keys):
    if not self.is_struct:
      return
    if not isinstance(keys, (list, tuple)):
      keys = [keys]
    for key in keys:
      if key not in self.params:
        raise KeyError('%s must be set in params' % key)

  def has_field(self, key):
    if self.is_struct:
      return key in self.params
    return self.params.HasField(key)
