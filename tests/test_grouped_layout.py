"""The grid of a grouped GEMM launch (er_gemm_grouped_layout / er_gemm_grouped_coords: host code of the library, the very
functions the launch uses): every (problem, tile, k-split) is computed by exactly one workgroup, and in the XCD region
all tiles of one k-split sit on ONE XCD (block % 8) - or, with 4 / 2 splits, on 2 / 4 neighbouring XCDs - the placement
the weight-gradient launch relies on for its operands to cross the HBM side once.  No device needed."""
import ctypes
import random

import pytest

I = ctypes.c_int32


def _layout(lib, tiles, splits, by_xcd):
  n = len(tiles)
  t, s = (I * n)(*tiles), (I * n)(*splits)
  flags = (I * n)(*by_xcd) if by_xcd is not None else None
  start, xstart, xsplits, xper = (I * (n + 1))(), (I * (n + 1))(), (I * n)(), (I * n)()
  grid = lib.er_gemm_grouped_layout(t, s, n, flags, start, xstart, xsplits, xper)
  assert grid >= 0
  return grid, t, start, xstart, xsplits, xper


def _coords(lib, lay, n, b):
  _, t, start, xstart, xsplits, xper = lay
  p, tile, split, plain = I(), I(), I(), I()
  rc = lib.er_gemm_grouped_coords(t, start, xstart, xsplits, xper, n, b, ctypes.byref(p), ctypes.byref(tile),
                                  ctypes.byref(split), ctypes.byref(plain))
  assert rc == 0
  return p.value, tile.value, split.value, plain.value


CASES = [
    ([40, 8, 2, 8, 8, 2], [8] * 6),          # DeepFM's weight gradients at B = 4096, stand-alone: 8 splits each
    ([40, 8, 2, 8, 8, 2], [4] * 6),          # ... in the step's tail: 4 splits, each on 2 XCDs
    ([4, 2, 1], [100, 100, 13]),             # 96 splits by XCD + 4 legacy, 8 + 5
    ([16] * 12, [4] * 12),
    ([1], [1]),
    ([3, 5, 7, 1], [16, 7, 2, 4]),           # odd tile counts under 2 / 4 splits: idle slots
]


def _expected_region(splits_p, flag):
  """(splits placed by XCD, XCDs per split)"""
  if not flag:
    return 0, 0
  if splits_p >= 8:
    return 8 * (splits_p // 8), 1
  if splits_p in (2, 4):
    return splits_p, 8 // splits_p
  return 0, 0


@pytest.mark.parametrize('mode', ['legacy', 'xcd', 'mixed'])
@pytest.mark.parametrize('case', range(len(CASES) + 20))
def test_every_tile_of_every_split_exactly_once(built_lib, case, mode):
  lib = ctypes.CDLL(built_lib)
  if case < len(CASES):
    tiles, splits = CASES[case]
  else:
    rng = random.Random(case)
    n = rng.randint(1, 16)
    tiles = [rng.randint(1, 50) for _ in range(n)]
    splits = [rng.choice([1, 2, 4, 7, 8, 9, 16, 24, 31, 64, 100]) for _ in range(n)]
  n = len(tiles)
  flags = {'legacy': None, 'xcd': [1] * n, 'mixed': [(case + p) % 2 for p in range(n)]}[mode]
  lay = _layout(lib, tiles, splits, flags)
  grid, _, start, xstart, xsplits, xper = lay
  work = sum(a * b for a, b in zip(tiles, splits))
  seen, idle = {}, [0] * n
  for b in range(grid):
    p, tile, split, plain = _coords(lib, lay, n, b)
    assert 0 <= p < n
    if split < 0:
      idle[p] += 1
      continue
    assert 0 <= tile < tiles[p] and 0 <= split < splits[p]
    assert (p, tile, split) not in seen, 'computed twice'
    seen[(p, tile, split)] = (b, plain)
  assert len(seen) == work and grid == work + sum(idle)
  for p in range(n):
    xs, xp = _expected_region(splits[p], flags[p] if flags else 0)
    assert (xsplits[p], xper[p]) == (xs, xp)
    # idle workgroups: only the last tile slot of a split shared by 2 / 4 XCDs, never a whole XCD's worth
    assert idle[p] == (((-tiles[p]) % xp) * splits[p] if xp > 1 else 0) and idle[p] < 8
  by_split = {}
  for (p, tile, split), (b, plain) in seen.items():
    xs, xp = _expected_region(splits[p], flags[p] if flags else 0)
    assert bool(plain) == (split < xs)
    if plain:
      by_split.setdefault((p, split), set()).add(b % 8)
      if xp == 1:
        assert b % 8 == split % 8            # XCD x holds splits x, x + 8, ...
      else:
        assert (b % 8) // xp == split        # split s on XCDs [s * xp, (s + 1) * xp)
  for (p, split), xcds in by_split.items():
    xp = max(1, xper[p])
    assert len(xcds) <= xp
  if flags is None:
    assert xstart[n] == 0 and grid == work


def test_deepfm_weight_gradients_keep_a_share_of_the_batch_per_xcd(built_lib):
  lib = ctypes.CDLL(built_lib)
  tiles = CASES[0][0]
  lay = _layout(lib, tiles, [8] * 6, [1] * 6)
  assert lay[0] == 544 and lay[2][len(tiles)] == 0
  for b in range(lay[0]):
    assert _coords(lib, lay, len(tiles), b)[2] == b % 8          # an eighth of the batch rows per XCD
  lay = _layout(lib, tiles, [4] * 6, [1] * 6)
  assert lay[0] == 272
  for b in range(lay[0]):
    assert _coords(lib, lay, len(tiles), b)[2] == (b % 8) // 2   # a quarter per XCD pair
