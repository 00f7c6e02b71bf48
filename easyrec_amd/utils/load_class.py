"""Class registries: the plugin lookups the kernels sit behind.

Mirror of reference easy_rec/python/utils/load_class.py:203-249: `get_register_class_meta` (models
and inputs register themselves by class name; `EasyRecModel.create_class('DeepFM')`) and
`load_keras_layer(name)` (backbone `keras_layer.class_name` -> easyrec_amd.layers.keras.<name>).
"""
import importlib
import logging
import pydoc
from abc import ABCMeta


def register_class(class_map, class_name, cls):
  assert class_name not in class_map or class_map[class_name] == cls, \
      'confilict class %s , %s is already register to be %s' % (cls, class_name, str(class_map[class_name]))
  logging.debug('register class %s' % class_name)
  class_map[class_name] = cls


def get_register_class_meta(class_map, have_abstract_class=True):

  class RegisterABCMeta(ABCMeta):

    def __new__(mcs, name, bases, attrs):
      newclass = super(RegisterABCMeta, mcs).__new__(mcs, name, bases, attrs)
      register_class(class_map, name, newclass)

      @classmethod
      def create_class(cls, name):
        if name in class_map:
          return class_map[name]
        raise Exception('Class %s is not registered. Available ones are %s' % (name, list(class_map.keys())))

      setattr(newclass, 'create_class', create_class)
      return newclass

  return RegisterABCMeta


def load_keras_layer(name):
  """(layer_class, is_customize).  Only this package's layers exist (no tf.keras fallback)."""
  name = (name or '').strip()
  if not name:
    return None
  cls = pydoc.locate('easyrec_amd.layers.keras.' + name)
  if cls is not None:
    return cls, True
  return None, False


def import_all_models():
  """Import every model module so that the registry is populated."""
  for mod in ('deepfm', 'dcn', 'multi_tower_din', 'mmoe', 'rank_model', 'multi_task_model', 'wide_and_deep', 'fm',
              'multi_tower', 'dlrm', 'simple_multi_task', 'ple', 'dbmtl'):
    try:
      importlib.import_module('easyrec_amd.model.' + mod)
    except ImportError as e:  # pragma: no cover
      logging.warning('model module %s not importable: %s', mod, e)
