"""CSVInput: separator-delimited text -> batches (reference easy_rec/python/input/csv_input.py:17-175).

`decode_csv` semantics used by the hot-path configs: split each line on `data_config.separator`,
empty cells take the field's default (utils/input_utils.py:11-36).  The decode itself is native, as in the
reference (tf.decode_csv is a C++ kernel): `er_decode_csv_host` splits and parses a whole batch in one pass, string
cells stay (begin, length) views of the file's bytes until they are packed for hashing (input.py PackedCol) - no
per-cell Python work on the Criteo layout.  The MI355X benchmark feeds device-resident synthetic batches.
"""
import logging

import numpy as np

from easyrec_amd.input.input import Input, get_type_defaults
from easyrec_amd.protos.dataset_pb2 import DatasetConfig


class CSVInput(Input):

  def __init__(self, data_config, feature_configs, input_path=None, task_index=0, task_num=1, **kwargs):
    super(CSVInput, self).__init__(data_config, feature_configs, input_path, **kwargs)
    self._with_header = data_config.with_header
    # one process per GPU: every worker reads its own part of the data (csv_input.py:109-127, input.py:1018-1023
    # `_safe_shard`): whole files when data_config.file_shard, else every task_num-th line
    if data_config.chief_redundant:
      task_index, task_num = max(task_index - 1, 0), max(task_num - 1, 1)
    self._task_index, self._task_num = int(task_index), int(task_num)
    self._line_no = 0  # data lines seen so far (line sharding runs over the concatenation of the files)
    import os
    self.native_decode = os.environ.get('EASYREC_AMD_NATIVE_CSV', '1') != '0'  # else the line-by-line Python path
    # host threads of the native decoder (0: one per hardware thread, at most 8; 1: the single-pass decoder)
    self.decode_threads = int(os.environ.get('EASYREC_AMD_CSV_THREADS', '0'))
    self._decode_buffers = {}

  @staticmethod
  def _split_quoted(line, sep):
    """tf.decode_csv(use_quote_delim=True) on one line that contains '"': a cell that STARTS with a quote runs to its
    closing quote (separators inside are data, '""' is one quote character); a quote inside an unquoted cell is an
    error, as in TensorFlow ("Unquoted fields cannot have quotes/CRLFs inside")."""
    parts, i, n = [], 0, len(line)
    while True:
      if i < n and line[i] == '"':
        cell, i = [], i + 1
        while True:
          j = line.find('"', i)
          if j < 0:
            raise ValueError('Quoted field has to end with quote followed by delim or end: %r' % line[:80])
          cell.append(line[i:j])
          if j + 1 < n and line[j + 1] == '"':  # escaped quote
            cell.append('"')
            i = j + 2
            continue
          i = j + 1
          break
        if i < n and not line.startswith(sep, i):
          raise ValueError('Quoted field has to end with quote followed by delim or end: %r' % line[:80])
        parts.append(''.join(cell))
      else:
        j = line.find(sep, i)
        cell = line[i:] if j < 0 else line[i:j]
        if '"' in cell:
          raise ValueError('Unquoted fields cannot have quotes/CRLFs inside: %r' % line[:80])
        parts.append(cell)
        i = n if j < 0 else j
      if i >= n:
        return parts
      i += len(sep)
      if i >= n:  # the line ends with a separator: one more, empty, cell
        parts.append('')
        return parts

  def _parse_lines(self, lines):
    sep = self._data_config.separator
    n_fields = len(self._input_fields)
    cols = [[] for _ in range(n_fields)]
    defaults = [get_type_defaults(t, v) for t, v in zip(self._input_field_types, self._input_field_defaults)]
    for line in lines:
      line = line.rstrip('\n').rstrip('\r')
      parts = self._split_quoted(line, sep) if '"' in line else line.split(sep)
      assert len(parts) == n_fields, 'expected %d fields, got %d: %r' % (n_fields, len(parts), line[:80])
      for i, p in enumerate(parts):
        if p == '':
          p = defaults[i]
        elif self._input_field_types[i] != DatasetConfig.STRING:
          t = self._input_field_types[i]
          p = int(p) if t in (DatasetConfig.INT32, DatasetConfig.INT64) else float(p)
        cols[i].append(p)
    return {name: cols[i] for i, name in enumerate(self._input_fields)}

  # -- worker sharding -------------------------------------------------------------------------------------------
  def _my_paths(self):
    paths = self._input_path if isinstance(self._input_path, list) else self._input_path.split(',')
    if self._task_num > 1 and self._data_config.file_shard:
      paths = [p for i, p in enumerate(paths) if i % self._task_num == self._task_index]
    return paths

  def _my_lines(self, data):
    """bytes of one file (header already removed) -> the bytes of this worker's lines, newline-terminated.  Blank
    lines are dropped; with line sharding line k of the data set belongs to worker k % task_num."""
    if not data.endswith(b'\n'):
      data += b'\n'
    if self._task_num == 1 or self._data_config.file_shard:
      return data
    text = np.frombuffer(data, dtype=np.uint8)
    ends = np.flatnonzero(text == 10) + 1            # one past every newline
    starts = np.concatenate([[0], ends[:-1]])
    body = ends - starts - 1 - (text[np.maximum(ends - 2, 0)] == 13)  # length without "\n" / "\r\n"
    keep = body > 0
    starts, ends = starts[keep], ends[keep]
    mine = (self._line_no + np.arange(len(starts))) % self._task_num == self._task_index
    self._line_no += len(starts)
    starts, ends = starts[mine], ends[mine]
    if len(starts) == 0:
      return b''
    length = ends - starts
    offs = np.zeros(len(length) + 1, dtype=np.int64)
    np.cumsum(length, out=offs[1:])
    src = np.repeat(starts - offs[:-1], length) + np.arange(int(offs[-1]), dtype=np.int64)
    return text[src].tobytes()

  def _read(self, path):
    with open(path, 'rb') as f:
      data = f.read()
    if self._with_header:
      data = data[data.index(b'\n') + 1:] if b'\n' in data else b''
    return self._my_lines(data)

  # -- native decode: one pass of er_decode_csv_host per batch ---------------------------------------------------
  def _native_ok(self):
    sep = self._data_config.separator
    return self.native_decode and len(sep.encode('utf-8')) == 1 and sep not in ('\n', '\r')

  def _columns_from_decoded(self, text, n, ints, flts, empty, begin, length):
    """Decoded arrays of n rows -> the {input_name: column} dict `preprocess` takes: numpy arrays for numeric fields
    (defaults filled in), PackedCol views of the text for string fields."""
    from easyrec_amd.input.input import PackedCol
    cols = {}
    for f, name in enumerate(self._input_fields):
      t = self._input_field_types[f]
      miss = empty[f, :n] != 0
      default = get_type_defaults(t, self._input_field_defaults[f])
      if t == DatasetConfig.STRING:
        if miss.any() and default not in ('', b''):
          col = PackedCol(text, begin[f, :n].copy(), length[f, :n].copy())
          cols[name] = [default if miss[i] else col[i] for i in range(n)]
        else:
          cols[name] = PackedCol(text, begin[f, :n].copy(), length[f, :n].copy())
      elif t in (DatasetConfig.INT32, DatasetConfig.INT64):
        cols[name] = np.where(miss, np.int64(default), ints[f, :n])
      else:
        cols[name] = np.where(miss, np.float64(default), flts[f, :n])
    return cols

  def _native_batches(self, paths, drop_remainder):
    from easyrec_amd import kernels
    be = kernels.hip()
    B = self._batch_size
    kinds = [0 if t == DatasetConfig.STRING else 1 if t in (DatasetConfig.INT32, DatasetConfig.INT64) else 2
             for t in self._input_field_types]
    sep = self._data_config.separator
    carry = None  # decoded rows of a batch that straddles two files: per-field python lists
    for path in paths:
      data = self._read(path)
      if b'"' in data:
        # quoted cells (tf.decode_csv's use_quote_delim): the zero-copy (begin, length) cells of the native decoder
        # cannot express an un-escaped '""'; such files take the line-by-line path (not the Criteo / Taobao layouts)
        if not getattr(self, '_warned_quoted', False):
          self._warned_quoted = True
          logging.warning('%s contains quote characters: read line by line in Python (~13x slower than the native decoder); '
                          'a quoted cell may not contain a line break', path)
        lines = [ln for ln in data.decode('utf-8').split('\n') if ln.strip('\r')]
        for ln in lines:
          if ln.count('"') % 2:
            raise ValueError('%s: a line with an unbalanced quote - quoted cells that span several lines are not '
                             'supported (tf.data.TextLineDataset splits on line breaks first as well): %r' % (path, ln[:80]))
        for i in range(0, len(lines), B):
          part = self._parse_lines(lines[i:i + B])
          if carry is None and len(lines[i:i + B]) == B:
            yield self.preprocess(part)
            continue
          carry = part if carry is None else {k: list(carry[k]) + list(part[k]) for k in part}
          n_have = len(next(iter(carry.values())))
          if n_have >= B:
            yield self.preprocess({k: v[:B] for k, v in carry.items()})
            carry = {k: v[B:] for k, v in carry.items()} if n_have > B else None
        continue
      text = np.frombuffer(data, dtype=np.uint8)
      pos = 0
      while pos < len(text):
        want = B - (len(next(iter(carry.values()))) if carry else 0)
        # (the decoded arrays are consumed - copied or turned into new arrays - by _columns_from_decoded before the next
        # call: they are reused across the batches of this reader)
        n, used, ints, flts, empty, begin, length = be.decode_csv_host(text[pos:], sep, kinds, want,
                                                                       threads=self.decode_threads, out=self._decode_buffers)
        if n == 0:
          break
        chunk = text[pos:]
        pos += used
        cols = self._columns_from_decoded(chunk, n, ints, flts, empty, begin, length)
        if carry is None and n == B:
          yield self.preprocess(cols)
          continue
        # a partial batch (file end) or its completion: fall back to plain lists for the few rows involved
        lists = {k: [v[i] for i in range(n)] for k, v in cols.items()}
        carry = lists if carry is None else {k: carry[k] + lists[k] for k in lists}
        if len(next(iter(carry.values()))) == B:
          yield self.preprocess(carry)
          carry = None
    if carry is not None and not drop_remainder:
      n = len(next(iter(carry.values())))
      yield self.preprocess({k: v + [v[-1]] * (B - n) for k, v in carry.items()})

  def batches(self, num_epochs=None, drop_remainder=True):
    """Yield batch dicts from the input file(s)."""
    if self._native_ok():
      for _ in range(num_epochs or self._data_config.num_epochs or 1):
        self._line_no = 0
        for b in self._native_batches(self._my_paths(), drop_remainder):
          yield b
      return
    epochs = num_epochs or self._data_config.num_epochs or 1
    B = self._batch_size
    for _ in range(epochs):
      buf = []
      self._line_no = 0
      for path in self._my_paths():
        for line in self._read(path).decode('utf-8').split('\n'):
          if not line.strip('\r'):
            continue
          buf.append(line)
          if len(buf) == B:
            yield self.preprocess(self._parse_lines(buf))
            buf = []
      if buf and not drop_remainder:
        while len(buf) < B:
          buf.append(buf[-1])
        yield self.preprocess(self._parse_lines(buf))
