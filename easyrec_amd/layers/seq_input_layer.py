"""SeqInputLayer: the key / history-sequence embeddings of a `seq_att_groups` entry (DIN, BST).

Mirror of reference easy_rec/python/layers/seq_input_layer.py:19-143:
  __call__(features, group_name) -> {'key': [B, sum E_key], 'hist_seq_emb': [B, L, sum E_hist],
                                     'hist_seq_len': [B], 'aux_hist_seq_emb_list': [...]}
  * it owns a SECOND FeatureColumnParser (:29-31), and looks keys up under variable_scope(group_name)
    (:58-74): the key of a DIN tower is a different table from the same feature in a plain tower
    unless `embedding_name` shares them;
  * the embedding regulariser is applied to every key output (:68-70); MultiTowerDIN applies it to the
    concatenated history output (model/multi_tower_din.py:54-60).
Sequence lookups keep the time axis (`EmbeddingColumn._get_sequence_dense_tensor`,
compat/feature_column/feature_column_v2.py:3616-3640): ids [B, L] padded with -1 -> [B*L, E] rows, zero
for padding; they run inside the model's single fused `er_emb_fwd` launch like every other lookup.

Padding: the lookups fill the static [B, max_seq_len] buffers; what the model gets is the slice up to the loaded
batch's longest sequence (`DeviceFeatures.seq_pad_len`), which is what the reference's sparse -> dense conversion
produces - BatchNorm inside the attention MLP normalises over exactly those positions.
Keys can be taken from the feature group the sequence features sit in (`feature_name_to_output_tensors`, :66-79).
"""
from collections import OrderedDict

from easyrec_amd.feature_column.feature_column import FeatureColumnParser
from easyrec_amd.layers.input_layer import declare_lookup
from easyrec_amd.protos.feature_config_pb2 import WideOrDeep


class SeqInputLayer(object):

  def __init__(self, feature_configs, feature_groups_config, embedding_regularizer=None, ev_params=None, engine=None):
    self._feature_groups_config = OrderedDict((x.group_name, x) for x in feature_groups_config)
    self._fc_parser = FeatureColumnParser(feature_configs, self.get_wide_deep_dict(), ev_params=ev_params)
    self._embedding_regularizer = embedding_regularizer
    assert engine is not None
    self._engine = engine
    self._plan = {}

  def get_wide_deep_dict(self):
    d = {}
    for cfg in self._feature_groups_config.values():
      for x in cfg.seq_att_map:
        for key in x.key:
          d[key] = WideOrDeep.DEEP
        for hist in x.hist_seq:
          d[hist] = WideOrDeep.DEEP
        for hist in x.aux_hist_seq:
          d[hist] = WideOrDeep.DEEP
    return d

  def _declare(self, features, group_name, scope_name, given_keys, allow_key_search):
    cfg = self._feature_groups_config[group_name]
    cols = dict(self._fc_parser.deep_columns)
    cols.update(self._fc_parser.sequence_columns)
    eng = self._engine
    B = eng.batch_size
    keys, hists, key_plan = [], [], []  # key_plan: ('own', column) | ('given', feature name), in config order
    for x in cfg.seq_att_map:
      assert len(x.aux_hist_seq) == 0, 'aux_hist_seq is outside the hot-path scope'
      for k in x.key:
        if k not in given_keys or (given_keys[k] is None and allow_key_search):
          keys.append(cols[k])
          key_plan.append(('own', cols[k]))
        else:
          assert given_keys[k] is not None, \
              'When allow_key_search is False, key: %s should defined in same feature group.' % k
          key_plan.append(('given', k))
      hists.extend(cols[h] for h in x.hist_seq)
    lens = {features.seqs[h.raw_name]['ids'].shape[1] for h in hists}
    assert len(lens) == 1, 'SequenceFeature Error: the history sequences of group %s differ in max length' % group_name
    L = lens.pop()
    kkey, hkey = 'seq:%s:%s:key' % (scope_name, group_name), 'seq:%s:%s:hist' % (scope_name, group_name)
    if keys:
      eng.declare_group(kkey, sum(c.dimension for c in keys), self._embedding_regularizer)
    col, own_cols = 0, {}
    for c in keys:
      declare_lookup(eng, features, c, scope_name, kkey, col, B)
      own_cols[id(c)] = (col, c.dimension)
      col += c.dimension
    eng.declare_seq_output(hkey, B * L, sum(c.dimension for c in hists), self._embedding_regularizer)
    col = 0
    for c in hists:
      declare_lookup(eng, features, c, scope_name, hkey, col, B * L, seq=True)
      col += c.dimension
    self._plan[(scope_name, group_name)] = dict(
        kkey=kkey if keys else None, hkey=hkey, L=L, len_name=hists[0].raw_name, key_plan=key_plan, own_cols=own_cols,
        all_own=all(kind == 'own' for kind, _ in key_plan), hist_width=sum(c.dimension for c in hists))

  def __call__(self, features, group_name, feature_name_to_output_tensors={}, allow_key_search=True,
               scope_name=None, requires_grad=True):
    import torch
    scope_name = scope_name or group_name
    given = feature_name_to_output_tensors or {}
    if (scope_name, group_name) not in self._plan:
      self._declare(features, group_name, scope_name, given, allow_key_search)
    p = self._plan[(scope_name, group_name)]
    eng = self._engine
    B = eng.batch_size
    own = None
    if not eng.finalized:
      own = eng.groups[p['kkey']]['out'] if p['kkey'] else None
      hist = eng.groups[p['hkey']]['out']
    else:
      eng.forward(features.version)
      own = eng.group_tensor(p['kkey'], requires_grad=requires_grad) if p['kkey'] else None
      hist = eng.group_tensor(p['hkey'], requires_grad=requires_grad)
    if p['all_own']:
      key = own
    else:  # some keys are the outputs the enclosing feature group already computed (seq_input_layer.py:66-79)
      parts = []
      for kind, ref in p['key_plan']:
        if kind == 'given':
          parts.append(given[ref])
        else:
          c0, d = p['own_cols'][id(ref)]
          parts.append(own[:, c0:c0 + d])
      key = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    hist3 = hist.view(B, p['L'], p['hist_width'])
    Lm = features.seq_pad_len(p['len_name'])
    if Lm < p['L']:
      hist3 = hist3[:, :Lm]  # the batch's longest sequence: what the reference's padded tensor holds
    return {
        'key': key,
        'hist_seq_emb': hist3,
        'hist_seq_len': features.seqs[p['len_name']]['len'],
        'aux_hist_seq_emb_list': [],
    }
