"""The per-step loss bars of the GPU-vs-oracle TRAJECTORY tests, in one place, and why they are what they are.

From identical parameters (step 0) the GPU path must match the oracle to north_star's 1e-4.  From step 1 on the two
implementations have applied Adam updates computed in different fp32 summation orders, and these small BatchNorm models
amplify a difference of one unit in the last place by many orders of magnitude within a few steps (Adam normalises
near-zero gradients to O(lr) updates; ReLU ties flip whole examples).  How much is MEASURED, on the CPU, by
tests/test_chaos_bars.py: the oracle against itself with every gradient perturbed by one fp32 ulp of the tensor's largest
element (a model of another summation order), over a grid of model and noise seeds.  That test fails when a bar below is
looser than 10 x the measured envelope (or the 1e-4 floor): the bars are held to the measurement, not the other way round.

bar[step] multiplies max(1, |oracle loss|)."""

FLOOR = 1e-4  # north_star's bar on logits / loss
SLACK = 10.0  # a bar may exceed the measured one-ulp envelope at its step by at most this factor

TRAJECTORY = {
    # tests/test_deepfm_gpu.py::test_trajectory_matches_oracle (B = 256, 5 steps)
    'deepfm_criteo_small': (1e-4, 1e-4, 2e-3, 2e-3, 2e-3),
    # tests/test_kv_embedding.py::test_hash_table_sequence_features_match_the_oracle_on_the_gpu (B = 48, 3 steps)
    'din_taobao_small': (1e-4, 1e-4, 1e-2),
}


def bar(config, step):
  b = TRAJECTORY[config]
  return b[min(step, len(b) - 1)]
