"""EasyRec config schema (the drop-in boundary).

`from easyrec_amd.protos import pipeline_pb2; pipeline_pb2.EasyRecConfig()`
works like the reference's generated modules
(reference easy_rec/python/protos/*.proto -> *_pb2.py via scripts/gen_proto.sh).

The message classes are built at import time from `easyrec_schema.desc`, a
serialized `FileDescriptorSet` produced by `tools/gen_schema.py` with this
package's own proto2 compiler (`proto_compiler.py`) - the same artefact `protoc
--descriptor_set_out` would emit.  Set `EASYREC_AMD_PROTO_ROOT` to a directory
containing `easy_rec/python/protos/*.proto` to compile the schema from source
at import time instead (e.g. to pick up new upstream fields).
"""
import os
import sys
import types

from google.protobuf import descriptor_pb2
from google.protobuf import descriptor_pool
from google.protobuf import message_factory
from google.protobuf import struct_pb2  # noqa: F401  (registers google/protobuf/struct.proto)

_HERE = os.path.dirname(os.path.abspath(__file__))
SCHEMA_FILE = os.path.join(_HERE, 'easyrec_schema.desc')
PROTO_PREFIX = 'easy_rec/python/protos/'

_pool = descriptor_pool.DescriptorPool()
_modules = {}


def _load_descriptor_set():
  root = os.environ.get('EASYREC_AMD_PROTO_ROOT')
  if root:
    from easyrec_amd.protos import proto_compiler
    pdir = os.path.join(root, PROTO_PREFIX)
    names = sorted(f for f in os.listdir(pdir) if f.endswith('.proto'))
    return proto_compiler.compile_protos([PROTO_PREFIX + n for n in names],
                                         root)
  if not os.path.exists(SCHEMA_FILE):
    raise ImportError(
        'easyrec_amd: schema descriptor %s is missing; run tools/gen_schema.py'
        % SCHEMA_FILE)
  fds = descriptor_pb2.FileDescriptorSet()
  with open(SCHEMA_FILE, 'rb') as fh:
    fds.ParseFromString(fh.read())
  return fds


def _register_wellknown(pool):
  for mod in (struct_pb2,):
    fdp = descriptor_pb2.FileDescriptorProto()
    mod.DESCRIPTOR.CopyToProto(fdp)
    try:
      pool.Add(fdp)
    except Exception:  # already present
      pass


def _build():
  _register_wellknown(_pool)
  fds = _load_descriptor_set()
  for fdp in fds.file:
    _pool.Add(fdp)
  for fdp in fds.file:
    fd = _pool.FindFileByName(fdp.name)
    base = os.path.basename(fdp.name)[:-len('.proto')]
    mod = types.ModuleType(__name__ + '.' + base + '_pb2')
    mod.DESCRIPTOR = fd
    for name, md in fd.message_types_by_name.items():
      setattr(mod, name, message_factory.GetMessageClass(md))
    for name, ed in fd.enum_types_by_name.items():
      wrapper = _EnumWrapper(ed)
      setattr(mod, name, wrapper)
      for v in ed.values:
        setattr(mod, v.name, v.number)
    _modules[base + '_pb2'] = mod
    sys.modules[mod.__name__] = mod
    globals()[base + '_pb2'] = mod


class _EnumWrapper(object):
  """Minimal stand-in for generated enum type wrappers (`Name`, `Value`)."""

  def __init__(self, enum_desc):
    self.DESCRIPTOR = enum_desc
    for v in enum_desc.values:
      setattr(self, v.name, v.number)

  def Name(self, number):
    return self.DESCRIPTOR.values_by_number[number].name

  def Value(self, name):
    return self.DESCRIPTOR.values_by_name[name].number

  def keys(self):
    return [v.name for v in self.DESCRIPTOR.values]

  def values(self):
    return [v.number for v in self.DESCRIPTOR.values]

  def items(self):
    return [(v.name, v.number) for v in self.DESCRIPTOR.values]


def get_pool():
  return _pool


def message_class(full_name):
  """`message_class('protos.DNN')` -> the python message class."""
  return message_factory.GetMessageClass(
      _pool.FindMessageTypeByName(full_name))


_build()
