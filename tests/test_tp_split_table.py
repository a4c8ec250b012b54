"""calc_split_type against the reference's own key table (model_files.h:145-190): the keys are read out of the reference
source when it is present (so a key added there and missing here fails), the expected classes are written down here."""
import os
import re

import pytest

import __graft_entry__ as ge

REF = "/root/reference/neural_speed/models/model_utils/model_files.h"

EXPECT = {
    # llama
    ".attention.wq.weight": "ROW", ".attention.wk.weight": "ROW", ".attention.wv.weight": "ROW",
    ".feed_forward.w1.weight": "ROW", ".feed_forward.w3.weight": "ROW",
    ".attention.wo.weight": "COLUMN", ".feed_forward.w2.weight": "COLUMN",
    # gpt-j
    ".attn.q_proj.weight": "ROW", ".attn.k_proj.weight": "ROW", ".attn.v_proj.weight": "ROW", ".mlp.fc_in.weight": "ROW",
    ".mlp.fc_in.bias": "COLUMN", ".mlp.fc_out.weight": "COLUMN", ".attn.out_proj.weight": "COLUMN",
    ".mlp.fc_out.bias": "ONLY_MASTER",
    # baichuan
    ".mlp.gate_proj.weight": "ROW", ".mlp.up_proj.weight": "ROW", ".self_attn.o_proj.weight": "COLUMN",
    ".mlp.down_proj.weight": "COLUMN", ".self_attn.W_pack.weight": "QKV_ROW",
    # chatglm2
    ".mlp.dense_h_to_4h.weight": "ROW", ".mlp.dense_4h_to_h.weight": "COLUMN", ".self_attention.dense.weight": "COLUMN",
    ".self_attention.query_key_value.weight": "QKV_ROW", ".self_attention.query_key_value.bias": "QKV_COLUMN",
}


def _par():
    ge.load_package()
    from neural_speed_amd import parallel as par
    return par


def test_every_key_of_the_table():
    par = _par()
    cls = {"ROW": par.TENSOR_1D_ROW, "COLUMN": par.TENSOR_1D_COLUMN, "QKV_ROW": par.TENSOR_1D_QKV_ROW,
           "QKV_COLUMN": par.TENSOR_1D_QKV_COLUMN, "ONLY_MASTER": par.TENSOR_1D_ONLY_MASTER}
    for key, want in EXPECT.items():
        assert par.calc_split_type("model.layers.7" + key) == cls[want], key
    for name in ("tok_embeddings.weight", "norm.weight", "output.weight", "layers.0.attention_norm.weight",
                 "model.layers.0.mlp.fc_in_not.weight"):
        assert par.calc_split_type(name) == par.TENSOR_NO_CHANGE, name


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree absent")
def test_no_key_of_the_reference_is_missing():
    src = open(REF).read()
    body = src[src.index("if (enable_tp) {"):src.index("void calc_ne()")]
    keys = set(re.findall(r'name\.find\("([^"]+)"\)', body))
    assert keys, "could not read the reference's key list"
    assert keys == set(EXPECT), sorted(keys ^ set(EXPECT))


def test_unsliceable_classes_are_refused_not_replicated():
    """a fused q|k|v tensor or a master-only bias is not a single N / K slice of a packed weight: shard_weight must
    raise instead of returning the tensor unchanged (a replicated K-split weight would be summed world times)"""
    par = _par()
    ctx = par.ParallelContext.__new__(par.ParallelContext)
    ctx.world, ctx.rank = 2, 0
    for t in (par.TENSOR_1D_QKV_ROW, par.TENSOR_1D_QKV_COLUMN, par.TENSOR_1D_ONLY_MASTER):
        with pytest.raises(NotImplementedError):
            ctx.shard_weight(object(), t)
