#!/usr/bin/env python
"""Compile the EasyRec config schema (.proto) into easyrec_amd/protos/easyrec_schema.desc.

Equivalent of the reference's scripts/gen_proto.sh (which downloads protoc 3.4.0 and emits
*_pb2.py) using this repo's own proto2 compiler; the output is a serialized
FileDescriptorSet, i.e. what `protoc --descriptor_set_out` writes.

  python tools/gen_schema.py [--proto_root /root/reference]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--proto_root', default='/root/reference')
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  # import the compiler without triggering easyrec_amd.protos.__init__ (needs the .desc)
  import importlib.util
  here = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location(
      'proto_compiler', os.path.join(here, '..', 'easyrec_amd', 'protos', 'proto_compiler.py'))
  pc = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(pc)
  prefix = 'easy_rec/python/protos/'
  pdir = os.path.join(args.proto_root, prefix)
  names = sorted(f for f in os.listdir(pdir) if f.endswith('.proto'))
  fds = pc.compile_protos([prefix + n for n in names], args.proto_root)
  out = args.out or os.path.join(here, '..', 'easyrec_amd', 'protos', 'easyrec_schema.desc')
  with open(out, 'wb') as fh:
    fh.write(fds.SerializeToString(deterministic=True))
  nmsg = sum(len(f.message_type) for f in fds.file)
  print('wrote %s: %d files, %d top-level messages, %d bytes' %
        (out, len(fds.file), nmsg, os.path.getsize(out)))


if __name__ == '__main__':
  main()
