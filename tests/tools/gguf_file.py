"""TEST INFRASTRUCTURE: writer of a minimal GGUF (v3) file the way the reference's reader consumes it
(/root/reference/neural_speed/models/model_utils/model_files.h:643-860 gguf_init_from_file, :862-986 gguf_load_from_file):
header {magic 'GGUF', version u32, n_tensors u64, n_kv u64}, key / value pairs (string = u64 length + bytes; the FIRST pair is
general.architecture), tensor infos {name, n_dims u32, ne u64[n_dims], type u32, offset u64 into the data section}, padding to
`general.alignment` (32), tensor data.  fp32 tensors only: the quantizer driver turns such a file into a BTLA-quantized one."""
import struct

import numpy as np

U32, F32, STRING, ARRAY = 4, 6, 8, 9   # gguf_type (gguf.h:120-135)
NAMES = {  # NE-converter name -> GGUF name (llama_utils.cpp:113-152 vs :165-204)
    "tok_embeddings.weight": "token_embd.weight", "norm.weight": "output_norm.weight", "output.weight": "output.weight",
    "attention_norm.weight": "attn_norm.weight", "attention.wq.weight": "attn_q.weight", "attention.wk.weight": "attn_k.weight",
    "attention.wv.weight": "attn_v.weight", "attention.wo.weight": "attn_output.weight", "ffn_norm.weight": "ffn_norm.weight",
    "feed_forward.w1.weight": "ffn_gate.weight", "feed_forward.w2.weight": "ffn_down.weight", "feed_forward.w3.weight": "ffn_up.weight"}


def gguf_name(ne_name):
    if ne_name.startswith("layers."):
        _, i, rest = ne_name.split(".", 2)
        return "blk.%s.%s" % (i, NAMES.get(rest, rest))   # the expert tensors (ffn_gate_inp, ffn_gate.N, ...) keep their names
    return NAMES[ne_name]


def ne_name(gname):
    inv = {v: k for k, v in NAMES.items()}
    if gname.startswith("blk."):
        _, i, rest = gname.split(".", 2)
        return "layers.%s.%s" % (i, inv.get(rest, rest))
    return inv[gname]


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<Q", len(b)) + b


def write_llama(path, hp, tensors):
    """hp: the hparams dict of ne_file.write; tensors: list of (NE-converter name, fp32 array)"""
    kv = [("general.architecture", STRING, "llama"), ("general.alignment", U32, 32), ("n_vocab", U32, hp["n_vocab"]),
          ("llama.embedding_length", U32, hp["n_embd"]), ("llama.attention.head_count", U32, hp["n_head"]),
          ("llama.attention.head_count_kv", U32, hp["n_head_kv"]), ("llama.block_count", U32, hp["n_layer"]),
          ("llama.rope.dimension_count", U32, hp["n_rot"]), ("llama.attention.layer_norm_rms_epsilon", F32, hp["norm_eps"]),
          ("llama.rope.freq_base", F32, hp["freq_base"]), ("ftype", U32, 0), ("llama.context_length", U32, hp["max_seq_len"]),
          ("llama.feed_forward_length", U32, hp["ffn_hidden_size"]), ("llama.expert_count", U32, hp.get("n_experts", 0)),
          ("llama.expert_used_count", U32, hp.get("n_experts_used", 0)), ("tokenizer.ggml.bos_token_id", U32, 1),
          ("tokenizer.ggml.eos_token_id", U32, 2), ("tokenizer.ggml.pad_token_id", U32, 0), ("tokenizer.ggml.sep_token_id", U32, 0)]
    out = bytearray(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(kv) + 2))
    for key, typ, val in kv:
        out += _s(key) + struct.pack("<I", typ)
        out += _s(val) if typ == STRING else struct.pack("<I" if typ == U32 else "<f", val)
    toks = [("<%d>" % i) for i in range(hp["n_vocab"])]
    out += _s("tokenizer.ggml.tokens") + struct.pack("<IIQ", ARRAY, STRING, len(toks)) + b"".join(_s(t) for t in toks)
    out += _s("tokenizer.ggml.scores") + struct.pack("<IIQ", ARRAY, F32, len(toks)) + np.zeros(len(toks), np.float32).tobytes()
    data = bytearray()
    for name, a in tensors:
        a = np.ascontiguousarray(a, np.float32)
        dims = tuple(reversed(a.shape))
        out += _s(gguf_name(name)) + struct.pack("<I", len(dims)) + struct.pack("<%dQ" % len(dims), *dims) + struct.pack("<IQ", 0, len(data))
        data += a.tobytes()
        data += b"\0" * (-len(data) & 31)
    out += b"\0" * (-len(out) & 31)
    with open(path, "wb") as f:
        f.write(out + data)
