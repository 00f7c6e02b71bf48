"""Background prefetch of host batches (the reference's `dataset.prefetch(prefetch_size)`, input/csv_input.py:165).

The decode (er_decode_csv_host), the string packing (numpy) and the host-to-device copy all release the GIL, so one
producer thread hides the input stage behind the device step; `transform` runs in the producer too (e.g.
`DeviceFeatures.pack` so that the consumer's load is a single copy)."""
import queue
import threading


class Prefetcher(object):
  """Iterate `source` in a background thread, at most `depth` items ahead.  Exceptions of the producer are re-raised
  in the consumer at the position where they occurred."""

  _END = object()

  def __init__(self, source, depth=2, transform=None):
    assert depth >= 1
    self._q = queue.Queue(maxsize=int(depth))
    self._stop = threading.Event()
    self._final = None  # terminal state seen by the consumer: ('end', None) or ('error', exception)
    self._thread = threading.Thread(target=self._run, args=(iter(source), transform), daemon=True)
    self._thread.start()

  def _put(self, item):
    while not self._stop.is_set():
      try:
        self._q.put(item, timeout=0.1)
        return True
      except queue.Full:
        continue
    return False

  def _run(self, it, transform):
    try:
      for x in it:
        if not self._put((None, transform(x) if transform is not None else x)):
          return
      self._put((None, self._END))
    except BaseException as e:  # noqa: BLE001  (handed to the consumer)
      self._put((e, None))

  def __iter__(self):
    return self

  def __next__(self):
    if self._final is not None:  # the producer is gone: repeat its last word instead of waiting on an empty queue
      if self._final[0] == 'error':
        raise self._final[1]
      raise StopIteration
    err, item = self._q.get()
    if err is not None:
      self._final = ('error', err)
      self.close()
      raise err
    if item is self._END:
      self._final = ('end', None)
      raise StopIteration
    return item

  def close(self):
    """Stop the producer (e.g. when the consumer leaves the loop early)."""
    self._stop.set()
    if self._final is None:
      self._final = ('end', None)  # closed early: later __next__ calls stop instead of blocking
    while True:
      try:
        self._q.get_nowait()
      except queue.Empty:
        break
