"""CSVInput: separator-delimited text -> batches (reference easy_rec/python/input/csv_input.py:17-175).

`decode_csv` semantics used by the hot-path configs: split each line on `data_config.separator`,
empty cells take the field's default (utils/input_utils.py:11-36).  Plumbing for config 1 (the
reference's own CPU-runnable case); the MI355X benchmark feeds device-resident synthetic batches.
"""
import numpy as np

from easyrec_amd.input.input import Input, get_type_defaults
from easyrec_amd.protos.dataset_pb2 import DatasetConfig


class CSVInput(Input):

  def __init__(self, data_config, feature_configs, input_path=None, **kwargs):
    super(CSVInput, self).__init__(data_config, feature_configs, input_path, **kwargs)
    self._with_header = data_config.with_header

  def _parse_lines(self, lines):
    sep = self._data_config.separator
    n_fields = len(self._input_fields)
    cols = [[] for _ in range(n_fields)]
    defaults = [get_type_defaults(t, v) for t, v in zip(self._input_field_types, self._input_field_defaults)]
    for line in lines:
      parts = line.rstrip('\n').rstrip('\r').split(sep)
      assert len(parts) == n_fields, 'expected %d fields, got %d: %r' % (n_fields, len(parts), line[:80])
      for i, p in enumerate(parts):
        if p == '':
          p = defaults[i]
        elif self._input_field_types[i] != DatasetConfig.STRING:
          t = self._input_field_types[i]
          p = int(p) if t in (DatasetConfig.INT32, DatasetConfig.INT64) else float(p)
        cols[i].append(p)
    return {name: cols[i] for i, name in enumerate(self._input_fields)}

  def batches(self, num_epochs=None, drop_remainder=True):
    """Yield batch dicts from the input file(s)."""
    paths = self._input_path if isinstance(self._input_path, list) else self._input_path.split(',')
    epochs = num_epochs or self._data_config.num_epochs or 1
    B = self._batch_size
    for _ in range(epochs):
      buf = []
      for path in paths:
        with open(path, 'r') as f:
          if self._with_header:
            next(f)
          for line in f:
            if not line.strip('\r\n'):
              continue
            buf.append(line)
            if len(buf) == B:
              yield self.preprocess(self._parse_lines(buf))
              buf = []
      if buf and not drop_remainder:
        while len(buf) < B:
          buf.append(buf[-1])
        yield self.preprocess(self._parse_lines(buf))
