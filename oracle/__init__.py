"""ORACLE - TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the hot path (alibaba/EasyRec v0.8.7 under
/root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; easyrec_amd/ never does, and the product path fails loudly without the HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * id hashing (farmhash_oracle.c): pinned to TF's published vectors - 4 x 64-bit Fingerprint64
    values quoted in TF's string_to_hash_bucket_op_test.py and the to_hash_bucket_fast docs example.
    Strings longer than 16 bytes: parity unpinned.
  * embedding lookup/combiners: pinned to the reference's own embed_test vectors
    (easy_rec/python/test/embed_test.py:22-151).
  * FM / cross / DIN / MMoE / DNN+BN / loss / Adam: the reference's tests hold no numeric
    expectation for them and TensorFlow cannot run here -> "parity unpinned": restated from the
    cited source lines + TF's documented op semantics, cross-checked fp32 vs fp64 and hand-written
    backward vs autograd.
"""
