"""CriteoInput: the Criteo terabyte set as three flat binary files per part.

Reference: easy_rec/python/input/criteo_binary_reader.py:15-210 (`BinaryDataset`: `*_label.bin` int32 [n],
`*_dense.bin` float32 [n, 13], `*_category.bin` uint32 [n, 26]; every worker takes an equal contiguous range of the
concatenated samples, a batch may straddle parts) and easy_rec/python/input/criteo_input.py:75-85 (columns
`f1..f13`, `c1..c26`, `label`, then the ordinary per-feature preprocessing).

The files are memory-mapped: a batch is three slices (two when it straddles a part boundary), widened and
transposed into the feature-major arrays the device buffers use.

Deviations (both are defects of the reference that only show when a batch straddles two parts, pinned by
tests/test_input_formats.py against vectors produced by the reference code itself):
  * `_get` (criteo_binary_reader.py:159-190) keeps `end_read_pos = start + batch_size` after moving on to the next
    part, so a straddling batch comes back with batch_size + (rows of the next part before that position) rows,
    which later batches deliver again.  Here a batch always has exactly batch_size rows - the FIRST batch_size rows
    of what the reference returns.
  * `_compute_global_start_pos` (criteo_binary_reader.py:87-92) advances `start_file_id` BEFORE adding the
    part's sample count, so with parts of unequal size a worker other than 0 can start in the wrong part.  The position
    arithmetic here is the intended one; for parts of equal size (and for worker 0 always) both agree.
"""
import glob

import numpy as np

from easyrec_amd.input.input import Input

N_DENSE, N_CATEGORY = 13, 26


class BinaryDataset(object):
  """Indexable batches (dense f32 [b, 13], category u32 [b, 26], label i32 [b, 1]) of one worker."""

  def __init__(self, label_bins, dense_bins, category_bins, batch_size=1, drop_last=False, global_rank=0,
               global_size=1):
    assert len(label_bins) == len(dense_bins) == len(category_bins) and label_bins
    self._labels = [np.memmap(p, dtype=np.int32, mode='r') for p in label_bins]
    self._dense = [np.memmap(p, dtype=np.float32, mode='r').reshape(-1, N_DENSE) for p in dense_bins]
    self._cats = [np.memmap(p, dtype=np.uint32, mode='r').reshape(-1, N_CATEGORY) for p in category_bins]
    counts = [len(x) for x in self._labels]
    for i, (d, c) in enumerate(zip(self._dense, self._cats)):
      assert len(d) == counts[i] and len(c) == counts[i], 'part %d: label / dense / category sizes differ' % i
    self._part_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    total = int(self._part_start[-1])
    # every worker gets the same number of samples (criteo_binary_reader.py:66-78)
    avg, res = total // global_size, total % global_size
    self._num_samples = avg + (1 if res > 0 else 0)
    if res > 0:
      self._start = (avg + 1) * global_rank if global_rank < res else avg * global_rank + res - 1
    else:
      self._start = avg * global_rank
    self._batch_size = batch_size
    self._num_entries = self._num_samples // batch_size
    self._last_batch_size = batch_size
    if not drop_last and self._num_samples % batch_size != 0:
      self._num_entries += 1
      self._last_batch_size = self._num_samples % batch_size
    self._total = total

  def __len__(self):
    return self._num_entries

  def _rows(self, arrays, begin, end):
    parts = []
    p = int(np.searchsorted(self._part_start, begin, side='right') - 1)
    while begin < end and p < len(arrays):
      lo = begin - int(self._part_start[p])
      hi = min(end, int(self._part_start[p + 1])) - int(self._part_start[p])
      parts.append(arrays[p][lo:hi])
      begin += hi - lo
      p += 1
    return parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0)

  def __getitem__(self, idx):
    if idx >= self._num_entries:
      raise IndexError()
    begin = self._start + idx * self._batch_size
    # like the reference every batch asks for batch_size rows; only the end of the data cuts it short
    end = min(begin + self._batch_size, self._total)
    return (np.asarray(self._rows(self._dense, begin, end)), np.asarray(self._rows(self._cats, begin, end)),
            np.asarray(self._rows(self._labels, begin, end)).reshape(-1, 1))


class CriteoInput(Input):

  def __init__(self, data_config, feature_configs, input_path=None, task_index=0, task_num=1, **kwargs):
    super(CriteoInput, self).__init__(data_config, feature_configs, input_path, **kwargs)
    labels, dense, cats = [], [], []
    if input_path is not None:
      # input_path: an object with label_path / dense_path / category_path glob lists (criteo_input.py:31-52), or a
      # dict with those keys
      get = (lambda k: input_path[k]) if isinstance(input_path, dict) else (lambda k: list(getattr(input_path, k)))
      lp, dp, cp = get('label_path'), get('dense_path'), get('category_path')
      assert len(lp) == len(dp) == len(cp), 'label_path_num(%d), dense_path_num(%d), category_path_num(%d) must be the ' \
          'same' % (len(lp), len(dp), len(cp))
      for a, b, c in zip(lp, dp, cp):
        la, db, cc = sorted(glob.glob(a)), sorted(glob.glob(b)), sorted(glob.glob(c))
        assert len(la) == len(db) == len(cc), 'label_path(%s) dense_path(%s) category_path(%s) matched different ' \
            'numbers of files (%d %d %d)' % (a, b, c, len(la), len(db), len(cc))
        labels += la
        dense += db
        cats += cc
    self._reader = BinaryDataset(labels, dense, cats, self._batch_size, drop_last=True, global_rank=task_index,
                                 global_size=task_num) if labels else None

  def to_columns(self, dense, category, labels):
    """criteo_input.py:75-85: columns f1..f13, c1..c26, label.  A data_config that names its 40 input fields
    differently (the reference's deepfm_on_criteo.config says label, F1.., C1..) is matched by position."""
    names = (['label'] + ['f%d' % (i + 1) for i in range(N_DENSE)] + ['c%d' % (i + 1) for i in range(N_CATEGORY)])
    if 'f1' not in self._input_fields and len(self._input_fields) == len(names):
      names = list(self._input_fields)
    cols = {names[0]: labels.reshape(-1)}
    cols.update({names[1 + i]: dense[:, i] for i in range(N_DENSE)})
    cols.update({names[1 + N_DENSE + i]: category[:, i].astype(np.int64) for i in range(N_CATEGORY)})
    return cols

  def batches(self, num_epochs=None, drop_remainder=True):
    assert drop_remainder, 'CriteoInput: the device buffers hold full batches only'
    for _ in range(num_epochs or self._data_config.num_epochs or 1):
      for i in range(len(self._reader)):
        dense, category, labels = self._reader[i]
        yield self.preprocess(self.to_columns(dense, category, labels))
