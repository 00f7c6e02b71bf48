"""A small proto2 compiler: `.proto` text -> `FileDescriptorProto`.

Why it exists: the drop-in boundary of this framework is EasyRec's protobuf
*text* config (`EasyRecConfig`, reference easy_rec/python/protos/pipeline.proto:14-61).
The reference generates `*_pb2.py` with a downloaded `protoc`
(reference scripts/gen_proto.sh:1-40); neither `protoc` nor `grpcio-tools`
exists in this image, but the protobuf *runtime* can build message classes from
descriptors.  This module is the missing front half: a tokenizer + recursive
descent parser for the proto2 subset used by the 46 schema files (syntax,
package, import, message, nested message/enum, oneof, optional/required/
repeated fields, scalar types, `[default = ...]`, `map<,>`, `option`,
`reserved`, `extensions`, stray `;`).

It is a compiler front end only; wire format, text format and reflection come
from `google.protobuf`.
"""
import os
import re

from google.protobuf import descriptor_pb2 as dpb

_FD = dpb.FieldDescriptorProto

_SCALARS = {
    'double': _FD.TYPE_DOUBLE,
    'float': _FD.TYPE_FLOAT,
    'int64': _FD.TYPE_INT64,
    'uint64': _FD.TYPE_UINT64,
    'int32': _FD.TYPE_INT32,
    'fixed64': _FD.TYPE_FIXED64,
    'fixed32': _FD.TYPE_FIXED32,
    'bool': _FD.TYPE_BOOL,
    'string': _FD.TYPE_STRING,
    'bytes': _FD.TYPE_BYTES,
    'uint32': _FD.TYPE_UINT32,
    'sfixed32': _FD.TYPE_SFIXED32,
    'sfixed64': _FD.TYPE_SFIXED64,
    'sint32': _FD.TYPE_SINT32,
    'sint64': _FD.TYPE_SINT64,
}

_LABELS = {
    'optional': _FD.LABEL_OPTIONAL,
    'required': _FD.LABEL_REQUIRED,
    'repeated': _FD.LABEL_REPEATED,
}

_TOKEN_RE = re.compile(
    r"""
    (?P<ws>\s+)
  | (?P<lc>//[^\n]*)
  | (?P<bc>/\*.*?\*/)
  | (?P<str>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')
  | (?P<num>[-+]?(?:0[xX][0-9a-fA-F]+|(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|inf|nan))
  | (?P<id>[A-Za-z_][A-Za-z0-9_.]*)
  | (?P<sym>[{}\[\]()<>=;,.:])
""", re.VERBOSE | re.DOTALL)


class ProtoSyntaxError(ValueError):
  pass


def _tokenize(text, fname):
  pos, out, line = 0, [], 1
  n = len(text)
  while pos < n:
    m = _TOKEN_RE.match(text, pos)
    if not m:
      raise ProtoSyntaxError('%s:%d: cannot tokenize near %r' %
                             (fname, line, text[pos:pos + 20]))
    kind = m.lastgroup
    tok = m.group(0)
    if kind not in ('ws', 'lc', 'bc'):
      out.append((kind, tok, line))
    line += tok.count('\n')
    pos = m.end()
  return out


_ESC = {
    'n': '\n', 't': '\t', 'r': '\r', '\\': '\\', "'": "'", '"': '"',
    'a': '\a', 'b': '\b', 'f': '\f', 'v': '\v', '?': '?'
}


def _unquote(tok):
  """Decode a proto string literal to python bytes-as-latin1 str."""
  body = tok[1:-1]
  out = []
  i = 0
  while i < len(body):
    c = body[i]
    if c != '\\':
      out.append(c)
      i += 1
      continue
    i += 1
    c = body[i]
    if c in _ESC:
      out.append(_ESC[c])
      i += 1
    elif c in 'xX':
      j = i + 1
      while j < len(body) and j < i + 3 and body[j] in '0123456789abcdefABCDEF':
        j += 1
      out.append(chr(int(body[i + 1:j], 16)))
      i = j
    elif c in '01234567':
      j = i
      while j < len(body) and j < i + 3 and body[j] in '01234567':
        j += 1
      out.append(chr(int(body[i:j], 8)))
      i = j
    else:
      out.append(c)
      i += 1
  return ''.join(out)


class _Parser(object):

  def __init__(self, text, fname):
    self.toks = _tokenize(text, fname)
    self.i = 0
    self.fname = fname

  # -- token helpers
  def peek(self):
    return self.toks[self.i] if self.i < len(self.toks) else ('eof', '', -1)

  def next(self):
    t = self.peek()
    self.i += 1
    return t

  def accept(self, val):
    if self.peek()[1] == val:
      self.i += 1
      return True
    return False

  def expect(self, val):
    t = self.next()
    if t[1] != val:
      raise ProtoSyntaxError('%s:%d: expected %r, got %r' %
                             (self.fname, t[2], val, t[1]))
    return t

  def ident(self):
    t = self.next()
    if t[0] != 'id':
      raise ProtoSyntaxError('%s:%d: expected identifier, got %r' %
                             (self.fname, t[2], t[1]))
    return t[1]

  def string(self):
    t = self.next()
    if t[0] != 'str':
      raise ProtoSyntaxError('%s:%d: expected string, got %r' %
                             (self.fname, t[2], t[1]))
    s = _unquote(t[1])
    while self.peek()[0] == 'str':  # adjacent literal concatenation
      s += _unquote(self.next()[1])
    return s

  def integer(self):
    t = self.next()
    if t[0] != 'num':
      raise ProtoSyntaxError('%s:%d: expected number, got %r' %
                             (self.fname, t[2], t[1]))
    return int(t[1], 0)

  # -- grammar
  def parse_file(self):
    fd = dpb.FileDescriptorProto()
    fd.name = self.fname
    while self.peek()[0] != 'eof':
      if self.accept(';'):
        continue
      kw = self.peek()[1]
      if kw == 'syntax':
        self.next()
        self.expect('=')
        syn = self.string()
        self.expect(';')
        if syn != 'proto2':
          fd.syntax = syn
      elif kw == 'package':
        self.next()
        fd.package = self.ident()
        self.expect(';')
      elif kw == 'import':
        self.next()
        if self.peek()[1] in ('public', 'weak'):
          self.next()
        fd.dependency.append(self.string())
        self.expect(';')
      elif kw == 'option':
        self.skip_option_stmt()
      elif kw == 'message':
        self.parse_message(fd.message_type.add())
      elif kw == 'enum':
        self.parse_enum(fd.enum_type.add())
      else:
        t = self.peek()
        raise ProtoSyntaxError('%s:%d: unexpected %r at file scope' %
                               (self.fname, t[2], t[1]))
    return fd

  def skip_option_stmt(self):
    self.expect('option')
    depth = 0
    while True:
      t = self.next()
      if t[0] == 'eof':
        raise ProtoSyntaxError('%s: unterminated option' % self.fname)
      if t[1] == '{':
        depth += 1
      elif t[1] == '}':
        depth -= 1
      elif t[1] == ';' and depth == 0:
        return

  def parse_enum(self, ed):
    self.expect('enum')
    ed.name = self.ident()
    self.expect('{')
    while not self.accept('}'):
      if self.accept(';'):
        continue
      if self.peek()[1] == 'option':
        self.skip_option_stmt()
        continue
      if self.peek()[1] == 'reserved':
        self.skip_to_semicolon()
        continue
      v = ed.value.add()
      v.name = self.ident()
      self.expect('=')
      v.number = self.integer()
      if self.accept('['):
        self.skip_brackets()
      self.expect(';')

  def skip_to_semicolon(self):
    while self.next()[1] != ';':
      pass

  def skip_brackets(self):
    depth = 1
    while depth:
      t = self.next()[1]
      if t == '[':
        depth += 1
      elif t == ']':
        depth -= 1

  def parse_message(self, md):
    self.expect('message')
    md.name = self.ident()
    self.parse_message_body(md)

  def parse_message_body(self, md):
    self.expect('{')
    while not self.accept('}'):
      if self.accept(';'):
        continue
      kw = self.peek()[1]
      if kw == 'message':
        self.parse_message(md.nested_type.add())
      elif kw == 'enum':
        self.parse_enum(md.enum_type.add())
      elif kw == 'oneof':
        self.next()
        od = md.oneof_decl.add()
        od.name = self.ident()
        idx = len(md.oneof_decl) - 1
        self.expect('{')
        while not self.accept('}'):
          if self.accept(';'):
            continue
          if self.peek()[1] == 'option':
            self.skip_option_stmt()
            continue
          f = self.parse_field(md, label=None)
          f.oneof_index = idx
      elif kw == 'option':
        self.skip_option_stmt()
      elif kw in ('reserved', 'extensions'):
        self.skip_to_semicolon()
      elif kw == 'map':
        self.parse_map_field(md)
      elif kw in _LABELS:
        self.next()
        if self.peek()[1] == 'group':
          raise ProtoSyntaxError('%s: groups are not supported' % self.fname)
        self.parse_field(md, label=_LABELS[kw])
      else:
        # proto3-style field without a label
        self.parse_field(md, label=None)

  def parse_field(self, md, label):
    f = md.field.add()
    f.label = label if label is not None else _FD.LABEL_OPTIONAL
    tname = self.ident()
    if tname in _SCALARS:
      f.type = _SCALARS[tname]
    else:
      f.type_name = tname  # resolved later (message or enum)
    f.name = self.ident()
    f.json_name = _json_name(f.name)
    self.expect('=')
    f.number = self.integer()
    if self.accept('['):
      self.parse_field_options(f)
    self.expect(';')
    return f

  def parse_field_options(self, f):
    while True:
      if self.accept('('):  # custom option, skip
        while self.next()[1] != ')':
          pass
        name = '()'
        while self.peek()[1] not in ('=',):
          self.next()
      else:
        name = self.ident()
      self.expect('=')
      t = self.next()
      if name == 'default':
        if t[0] == 'str':
          s = _unquote(t[1])
          while self.peek()[0] == 'str':
            s += _unquote(self.next()[1])
          if f.HasField('type') and f.type == _FD.TYPE_BYTES:
            s = ''.join(
                c if 32 <= ord(c) < 127 and c not in '\\"\'' else
                '\\%03o' % ord(c) for c in s)
          f.default_value = s
        else:
          f.default_value = self._normalize_default(f, t[1])
      elif name == 'packed':
        f.options.packed = (t[1] == 'true')
      elif name == 'deprecated':
        f.options.deprecated = (t[1] == 'true')
      if self.accept(','):
        continue
      self.expect(']')
      return

  @staticmethod
  def _normalize_default(f, tok):
    if not f.HasField('type'):  # enum value name (type resolved later)
      return tok
    if f.type in (_FD.TYPE_FLOAT, _FD.TYPE_DOUBLE):
      if tok in ('inf', '-inf', 'nan'):
        return tok
      v = float(tok)
      r = repr(v)
      return r
    if f.type == _FD.TYPE_BOOL:
      return tok
    if f.type_name:  # enum value name
      return tok
    return str(int(tok, 0))

  def parse_map_field(self, md):
    self.expect('map')
    self.expect('<')
    ktype = self.ident()
    self.expect(',')
    vtype = self.ident()
    self.expect('>')
    name = self.ident()
    self.expect('=')
    number = self.integer()
    if self.accept('['):
      self.skip_brackets()
    self.expect(';')
    entry = md.nested_type.add()
    entry.name = ''.join(p.capitalize() for p in name.split('_')) + 'Entry'
    entry.options.map_entry = True
    for i, (fname, tname) in enumerate((('key', ktype), ('value', vtype))):
      ef = entry.field.add()
      ef.name = fname
      ef.json_name = fname
      ef.number = i + 1
      ef.label = _FD.LABEL_OPTIONAL
      if tname in _SCALARS:
        ef.type = _SCALARS[tname]
      else:
        ef.type_name = tname
    f = md.field.add()
    f.name = name
    f.json_name = _json_name(name)
    f.number = number
    f.label = _FD.LABEL_REPEATED
    f.type_name = entry.name


def _json_name(name):
  parts = name.split('_')
  return parts[0] + ''.join(p[:1].upper() + p[1:] for p in parts[1:])


def parse_proto_text(text, fname):
  """Parse one `.proto` source into an (unresolved) FileDescriptorProto."""
  return _Parser(text, fname).parse_file()


# ---------------------------------------------------------------------------
# type resolution across a set of files
# ---------------------------------------------------------------------------


def _collect_symbols(fd, table):
  pkg = fd.package

  def walk(prefix, msgs, enums):
    for e in enums:
      table[prefix + '.' + e.name if prefix else e.name] = 'enum'
    for m in msgs:
      full = prefix + '.' + m.name if prefix else m.name
      table[full] = 'message'
      walk(full, m.nested_type, m.enum_type)

  walk(pkg, fd.message_type, fd.enum_type)


def _resolve(fd, table):
  pkg = fd.package

  def resolve_name(name, scope):
    if name.startswith('.'):
      if name[1:] in table:
        return name[1:]
      raise ProtoSyntaxError('%s: unknown type %s' % (fd.name, name))
    parts = scope.split('.') if scope else []
    for k in range(len(parts), -1, -1):
      cand_scope = '.'.join(parts[:k])
      cand = (cand_scope + '.' + name) if cand_scope else name
      if cand in table:
        return cand
    raise ProtoSyntaxError('%s: unknown type %r in scope %r' %
                           (fd.name, name, scope))

  def walk(prefix, msgs):
    for m in msgs:
      full = prefix + '.' + m.name if prefix else m.name
      for f in m.field:
        if f.type_name and not f.HasField('type'):
          target = resolve_name(f.type_name, full)
          f.type_name = '.' + target
          f.type = (_FD.TYPE_ENUM
                    if table[target] == 'enum' else _FD.TYPE_MESSAGE)
      walk(full, m.nested_type)

  walk(pkg, fd.message_type)


def compile_protos(paths, include_root, well_known=()):
  """Compile `.proto` files (paths relative to `include_root`) plus imports.

  Returns a `FileDescriptorSet` in dependency order.  Imports of
  `google/protobuf/*.proto` are satisfied from the runtime's own descriptors and
  are not re-emitted.
  """
  parsed = {}
  order = []

  def load(rel):
    if rel in parsed or rel.startswith('google/protobuf/'):
      return
    with open(os.path.join(include_root, rel), 'r') as fh:
      fd = parse_proto_text(fh.read(), rel)
    parsed[rel] = fd
    for dep in fd.dependency:
      load(dep)
    order.append(rel)

  for p in paths:
    load(p)

  table = {}
  for fd in parsed.values():
    _collect_symbols(fd, table)
  # well-known types reachable through google/protobuf imports
  from google.protobuf import struct_pb2, any_pb2, timestamp_pb2, duration_pb2, wrappers_pb2
  for mod in (struct_pb2, any_pb2, timestamp_pb2, duration_pb2, wrappers_pb2):
    wk = dpb.FileDescriptorProto()
    mod.DESCRIPTOR.CopyToProto(wk)
    _collect_symbols(wk, table)
  for fd in parsed.values():
    _resolve(fd, table)

  fds = dpb.FileDescriptorSet()
  for rel in order:
    fds.file.add().CopyFrom(parsed[rel])
  return fds
