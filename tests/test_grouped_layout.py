"""The grid of a grouped GEMM launch (er_gemm_grouped_layout / er_gemm_grouped_coords: host code of the library, the very
functions the launch uses): every (problem, tile, k-split) is computed by exactly one workgroup, and in the XCD region
all tiles of one k-split sit on ONE XCD (block % 8) - the placement the weight-gradient launch relies on for its
operands to cross the HBM side once.  No device needed."""
import ctypes
import random

import pytest


def _layout(lib, tiles, splits, by_xcd):
  n = len(tiles)
  I = ctypes.c_int32
  t, s = (I * n)(*tiles), (I * n)(*splits)
  start, xstart, xsplits = (I * (n + 1))(), (I * (n + 1))(), (I * n)()
  grid = lib.er_gemm_grouped_layout(t, s, n, int(by_xcd), start, xstart, xsplits)
  assert grid >= 0
  return grid, t, start, xstart, xsplits


def _coords(lib, t, start, xstart, xsplits, n, b):
  I = ctypes.c_int32
  p, tile, split, plain = I(), I(), I(), I()
  rc = lib.er_gemm_grouped_coords(t, start, xstart, xsplits, n, b, ctypes.byref(p), ctypes.byref(tile), ctypes.byref(split),
                                  ctypes.byref(plain))
  assert rc == 0
  return p.value, tile.value, split.value, plain.value


CASES = [
    ([40, 8, 2, 8, 8, 2], [8] * 6),          # DeepFM's weight gradients at B = 4096: 8 splits each, all by XCD
    ([4, 2, 1], [100, 100, 13]),             # DIN-like: 96 splits by XCD + 4 legacy, 8 + 5
    ([16] * 12, [4] * 12),                   # MMoE-like: fewer than 8 splits, legacy region only
    ([1], [1]),
    ([3, 5], [16, 7]),
]


@pytest.mark.parametrize('by_xcd', [0, 1])
@pytest.mark.parametrize('case', range(len(CASES) + 20))
def test_every_tile_of_every_split_exactly_once(built_lib, case, by_xcd):
  lib = ctypes.CDLL(built_lib)
  if case < len(CASES):
    tiles, splits = CASES[case]
  else:
    rng = random.Random(case)
    n = rng.randint(1, 16)
    tiles = [rng.randint(1, 50) for _ in range(n)]
    splits = [rng.choice([1, 2, 4, 7, 8, 9, 16, 24, 31, 64, 100]) for _ in range(n)]
  n = len(tiles)
  grid, t, start, xstart, xsplits = _layout(lib, tiles, splits, by_xcd)
  assert grid == sum(a * b for a, b in zip(tiles, splits))  # exact: no surplus workgroups
  seen = {}
  for b in range(grid):
    p, tile, split, plain = _coords(lib, t, start, xstart, xsplits, n, b)
    assert 0 <= p < n and 0 <= tile < tiles[p] and 0 <= split < splits[p]
    assert (p, tile, split) not in seen, 'computed twice'
    seen[(p, tile, split)] = (b, plain)
  assert len(seen) == grid
  xcd_of = {}
  for (p, tile, split), (b, plain) in seen.items():
    want_plain = bool(by_xcd) and split < 8 * (splits[p] // 8)
    assert bool(plain) == want_plain
    if plain:
      assert b % 8 == split % 8            # XCD x holds splits x, x + 8, ...
      xcd_of.setdefault((p, split), set()).add(b % 8)
  assert all(len(v) == 1 for v in xcd_of.values())
  if by_xcd:
    assert list(xsplits) == [8 * (s // 8) for s in splits]
  else:
    assert not any(xsplits) and xstart[n] == 0


def test_deepfm_weight_gradients_keep_an_eighth_of_the_batch_per_xcd(built_lib):
  lib = ctypes.CDLL(built_lib)
  tiles, splits = CASES[0]
  grid, t, start, xstart, xsplits = _layout(lib, tiles, splits, 1)
  assert grid == 544 and start[len(tiles)] == 0
  for b in range(grid):
    _, _, split, _ = _coords(lib, t, start, xstart, xsplits, len(tiles), b)
    assert split == b % 8
