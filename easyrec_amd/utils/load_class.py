"""Class registries: the plugin lookups the kernels sit behind.

API of reference easy_rec/python/utils/load_class.py:193-249: `get_register_class_meta(class_map)` gives the metaclass
through which models register themselves under their class name and gain `create_class(name)`
(`EasyRecModel.create_class('DeepFM')`); `load_keras_layer(name)` resolves a backbone `keras_layer.class_name`.

Here the registry is a small object (`ClassRegistry`) and the metaclass is a thin hook that records each new class
in it; duplicates are rejected unless the very same class object registers twice (module re-import).
"""
import importlib
import logging
from abc import ABCMeta


class ClassRegistry(object):
  """name -> class, backed by the caller's dict so existing references to the map keep working."""

  def __init__(self, backing):
    self._map = backing

  def add(self, name, cls):
    known = self._map.get(name)
    if known is not None and known is not cls:
      raise AssertionError('class name %s is taken: %r is registered, %r tried to register' % (name, known, cls))
    self._map[name] = cls
    logging.debug('registered %s -> %r', name, cls)

  def find(self, name):
    try:
      return self._map[name]
    except KeyError:
      raise Exception('Class %s is not registered. Available ones are %s' % (name, sorted(self._map)))


def register_class(class_map, class_name, cls):
  ClassRegistry(class_map).add(class_name, cls)


def get_register_class_meta(class_map, have_abstract_class=True):
  """A metaclass (ABCMeta-derived, so `@abstractmethod` works on the bases) that files every class created with it
  in `class_map` and gives it a `create_class(name)` classmethod looking names up in the same map."""
  registry = ClassRegistry(class_map)

  def _init(cls, name, bases, attrs):
    ABCMeta.__init__(cls, name, bases, attrs)
    registry.add(name, cls)
    cls.create_class = classmethod(lambda _cls, wanted: registry.find(wanted))

  return type('RegisterABCMeta', (ABCMeta,), {'__init__': _init})


def load_keras_layer(name):
  """(layer_class, is_customize) for a backbone `keras_layer.class_name`.  Only this package's layers exist (there is
  no tf.keras to fall back to): unknown names give (None, False)."""
  name = (name or '').strip()
  if not name:
    return None
  module = importlib.import_module('easyrec_amd.layers.keras')
  cls = getattr(module, name, None)
  if not isinstance(cls, type):
    return None, False
  # (False: a standard Keras layer, constructed from keyword arguments - layers/keras/standard.py)
  return cls, not getattr(cls, 'standard', False)


MODEL_MODULES = ('deepfm', 'dcn', 'multi_tower_din', 'mmoe', 'rank_model', 'multi_task_model', 'wide_and_deep', 'fm',
                 'multi_tower', 'dlrm', 'simple_multi_task', 'ple', 'dbmtl')


def import_all_models():
  """Import every model module so that the registry is populated."""
  for mod in MODEL_MODULES:
    try:
      importlib.import_module('easyrec_amd.model.' + mod)
    except ImportError as e:  # pragma: no cover
      logging.warning('model module %s not importable: %s', mod, e)
