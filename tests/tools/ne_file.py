"""TEST INFRASTRUCTURE: writer / reader of neural-speed's NE model file format (the non-GGUF container its converters emit
and /root/reference/neural_speed/models/model_utils/model_files.h reads: magic + version :1028-1062, 26 hparams
:1080-1145, vocab :1147-1176, tensor records :1177-1235 — header, name, padding to 32 bytes, data; a BTLA tensor's data is
the serialized blob, whose first 8 bytes are its size).  Used to hand the reference's UNCHANGED loader a synthetic model."""
import struct

import numpy as np

MAGIC_GGJT, VERSION = 0x67676A74, 3           # model_utils.h:40-47
NE_TYPE_F32, NE_TYPE_Q4_0, NE_TYPE_BTLA = 0, 2, 19   # core/data_types.h:32-55
NE_FTYPE_ALL_F32, NE_FTYPE_MOSTLY_Q_BTLA = 0, 10

HPARAMS = [  # (name, struct code) in file order (model_files.h:1080-1145)
    ("n_vocab", "I"), ("n_embd", "I"), ("n_mult", "I"), ("n_head", "I"), ("n_head_kv", "I"), ("n_layer", "I"), ("n_rot", "I"),
    ("ftype", "I"), ("max_seq_len", "I"), ("alibi_bias_max", "f"), ("clip_qkv", "f"), ("par_res", "I"),
    ("word_embed_proj_dim", "I"), ("do_layer_norm_before", "I"), ("multi_query_group_num", "I"), ("ffn_hidden_size", "I"),
    ("inner_hidden_size", "I"), ("n_experts", "I"), ("n_experts_used", "I"), ("n_embd_head_k", "I"), ("norm_eps", "f"),
    ("freq_base", "f"), ("freq_scale", "f"), ("rope_scaling_factor", "f"), ("original_max_position_embeddings", "I"),
    ("use_yarn", "I")]


def write(path, hparams, tensors, vocab_ids=(1, 2, 0, 0)):
    """tensors: list of (name, array): fp32 arrays ([N][K] or [K]) are written as NE_TYPE_F32; a (blob uint8, n, k) tuple
    as NE_TYPE_BTLA with ne = {k, n}."""
    with open(path, "wb") as f:
        f.write(struct.pack("<II", MAGIC_GGJT, VERSION))
        for name, code in HPARAMS:
            f.write(struct.pack("<" + code, hparams.get(name, 0)))
        f.write(struct.pack("<4i", *vocab_ids))   # bos, eos, pad, sep
        for i in range(hparams["n_vocab"]):
            word = ("<%d>" % i).encode()
            f.write(struct.pack("<I", len(word)) + word + struct.pack("<f", 0.0))
        for name, t in tensors:
            nb = name.encode()
            if isinstance(t, tuple) and isinstance(t[0], str) and t[0] == "q4_0":   # ("q4_0", fp32 [N][K] array): ggml Q4_0 blocks
                a = np.ascontiguousarray(t[1], np.float32)
                dims, typ, data = tuple(reversed(a.shape)), NE_TYPE_Q4_0, quant_q4_0(a)
            elif isinstance(t, tuple):
                blob, n, k = t
                dims, typ, data = (k, n), NE_TYPE_BTLA, bytes(memoryview(np.ascontiguousarray(blob)))
            else:
                a = np.ascontiguousarray(t, np.float32)
                dims, typ, data = tuple(reversed(a.shape)), NE_TYPE_F32, a.tobytes()
            f.write(struct.pack("<III", len(dims), len(nb), typ))
            f.write(struct.pack("<%dI" % len(dims), *dims))
            f.write(nb)
            f.write(b"\0" * (-f.tell() & 31))
            f.write(data)


def read(path):
    """-> (hparams dict, {name: (type, ne tuple, bytes)})"""
    with open(path, "rb") as f:
        buf = f.read()
    magic, version = struct.unpack_from("<II", buf, 0)
    assert (magic, version) == (MAGIC_GGJT, VERSION)
    off = 8
    hp = {}
    for name, code in HPARAMS:
        (hp[name],) = struct.unpack_from("<" + code, buf, off)
        off += 4
    off += 16
    for _ in range(hp["n_vocab"]):
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4 + ln + 4
    tensors = {}
    while off < len(buf):
        n_dims, name_len, typ = struct.unpack_from("<III", buf, off)
        off += 12
        ne = struct.unpack_from("<%dI" % n_dims, buf, off)
        off += 4 * n_dims
        name = buf[off:off + name_len].decode()
        off += name_len
        off += -off & 31
        if typ == NE_TYPE_BTLA:
            (size,) = struct.unpack_from("<Q", buf, off)
        elif typ == NE_TYPE_Q4_0:
            size = int(np.prod(ne)) // 32 * 18   # block_q4_0: fp16 d + 16 bytes of nibbles per 32 elements
        else:
            assert typ == NE_TYPE_F32, (name, typ)
            size = 4 * int(np.prod(ne))
        tensors[name] = (typ, ne, buf[off:off + size])
        off += size
    return hp, tensors


def quant_q4_0(a):
    """ggml's Q4_0 (the reference's own ne_quantize_q4_0, what its converter / quantizer makes of the token embedding):
    blocks of 32 along a row: fp16 d = (value of largest magnitude) / -8, codes clamp(round(x / d) + 8, 0, 15), element j and
    j + 16 share byte j"""
    x = a.reshape(-1, 32)
    idx = np.abs(x).argmax(1)
    mx = x[np.arange(x.shape[0]), idx]
    d = (mx / -8.0).astype(np.float32)
    inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
    q = np.clip(np.floor(x * inv[:, None] + 8.5), 0, 15).astype(np.uint8)
    out = np.zeros((x.shape[0], 18), np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.tobytes()


def dequant_q4_0(data, ne):
    """vectors/cpu/quantize.h:686-704: element j of a block is the low nibble of byte j, element j + 16 the high nibble,
    both minus 8, times the block's fp16 d.  -> fp64 [ne[1]][ne[0]]"""
    blk = np.frombuffer(data, np.uint8).reshape(-1, 18)
    d = blk[:, :2].copy().view(np.float16).astype(np.float64)            # [nblk][1]
    q = blk[:, 2:]
    vals = np.concatenate([(q & 15).astype(np.int32) - 8, (q >> 4).astype(np.int32) - 8], axis=1) * d
    return vals.reshape(ne[1], ne[0])
