"""TEST INFRASTRUCTURE ONLY - numpy restatement of the on-device batch assembly and dynamic masking
(stonkgs_amd/csrc/data.hip: stonk_mlm_mask, stonk_assemble_rows). Imported by tests/ only; the product never routes
through it.

What it restates: the reference's masking semantics (ref:src/stonkgs/data/indra_for_pretraining.py:33-77 - exactly
int(len * 0.15) positions per padded half, 80 / 10 / 10 split, labels = original id / -100), its row assembly
(:190-239) and its negative NSP pairing (:80-126), driven by the kernels' counter-based random stream instead of Python's
Mersenne Twister. Integer work: the GPU tests demand equality with these functions. Parity with the REFERENCE is pinned
where it can be - the distributional properties below and the bit-exact host path (tests/golden/masking.npz, produced by
the reference itself, against stonkgs_amd/data.py) - the counter-based stream itself has no reference counterpart."""
import numpy as np

K_SEED, K_SEED_ADD, K_ROW, K_POS = 0x9E3779B9, 0x7F4A7C15, 0x85EBCA77, 0x9E3779B1
K_D1, K_D2, K_D3, K_NEG, K_PART = 0x68E31DA4, 0xB5297A4D, 0x1B56C4E9, 0x2545F491, 0x632BE5AB
M32 = np.uint64(0xFFFFFFFF)


def hash32(x):
    """lowbias32 on uint32 arrays (stonk_hash32 in csrc/common.h)."""
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def _mul32(a, b):
    return (np.asarray(a, dtype=np.uint64) * np.uint64(b)) & M32


def base_key(seed):
    return hash32((_mul32(seed, K_SEED) + np.uint64(K_SEED_ADD)) & M32)


def mlm_mask(ids, half, vocab_text, vocab_ent, mask_id=103, k_text=None, k_ent=None, seed=0):
    ids = np.asarray(ids, dtype=np.int64)
    B, S = ids.shape
    assert S == 2 * half
    k_text = int(half * 0.15) if k_text is None else k_text
    k_ent = int(half * 0.15) if k_ent is None else k_ent
    out = ids.copy()
    labels = [np.full((B, half), -100, dtype=np.int64), np.full((B, half), -100, dtype=np.int64)]
    pos = np.arange(half, dtype=np.uint64)
    base = base_key(seed)
    for b in range(B):
        for h in range(2):
            rowkey = hash32(base ^ _mul32(b * 2 + h, K_ROW))
            key = hash32((rowkey + _mul32(pos, K_POS)) & M32)
            order = np.lexsort((np.arange(half), key))          # by key, ties by position
            chosen = order[: (k_text if h == 0 else k_ent)]
            vocab = vocab_text if h == 0 else vocab_ent
            for p in chosen:
                tok = ids[b, h * half + p]
                kk = key[p]
                if (hash32(kk ^ np.uint64(K_D1)) >> np.uint64(8)) < 13421773:
                    new = mask_id
                elif (hash32(kk ^ np.uint64(K_D2)) >> np.uint64(8)) < 8388608:
                    new = tok
                else:
                    new = int((hash32(kk ^ np.uint64(K_D3)) * np.uint64(vocab)) >> np.uint64(32))
                out[b, h * half + p] = new
                labels[h][b, p] = tok
    return out, labels[0], labels[1]


def assemble_rows(text_ids, text_attention, source, target, walks, sep_id=102, negative_rate=0.2, seed=0):
    text_ids, text_attention = np.asarray(text_ids, dtype=np.int64), np.asarray(text_attention, dtype=np.int64)
    walks = np.asarray(walks, dtype=np.int64)
    B, half = text_ids.shape
    W = walks.shape[1]
    assert 2 * W + 2 == half
    thr = int(negative_rate * 4294967296.0 + 0.5)
    base = base_key(seed)
    ids = np.zeros((B, 2 * half), dtype=np.int64)
    att = np.ones((B, 2 * half), dtype=np.int64)
    typ = np.concatenate([np.zeros((B, half), dtype=np.int64), np.ones((B, half), dtype=np.int64)], axis=1)
    nsp = np.zeros(B, dtype=np.int64)
    for b in range(B):
        neg = B > 1 and int(hash32(base ^ _mul32(b, K_NEG))) < thr
        row = b
        if neg:
            row = int((hash32(base ^ _mul32(b, K_PART)) * np.uint64(B)) >> np.uint64(32))
            if row == b:
                row = (b + 1) % B
        nsp[b] = 1 if neg else 0
        ids[b, :half] = text_ids[b]
        att[b, :half] = text_attention[b]
        ids[b, half:] = np.concatenate([walks[source[row]], [sep_id], walks[target[row]], [sep_id]])
    return ids, att, typ, nsp


def unpad_plan(attention_mask, text_labels=None, ent_labels=None, read=False):
    """numpy restatement of stonkgs_amd/csrc/unpad.hip (stonk_unpad_plan): which padded positions the trainable encoder
    keeps. Kept: attention_mask != 0, position 0 (the pooler's input, hf:modeling_bert.py:457-463),
    a labelled position of either half (the reference labels 15 % of the PADDED half,
    ref:src/stonkgs/data/indra_for_pretraining.py:33-77, and nn.CrossEntropyLoss reads those rows,
    ref:src/stonkgs/models/stonkgs_model.py:229-240); a sequence without any unmasked key keeps everything (the reference
    then attends uniformly over all S keys, hf:modeling_bert.py:111-136 with every score at finfo.min).
    Returns row_of_pos [B*S], pos_of_row [B*S], seq_offsets [B+1], row_mask [B*S]; with `read=True` also the READ rows
    (labelled positions and position 0 - the only rows whose last-layer output the heads read): read_rows [B*S],
    read_of_pos [B*S], read_offsets [B+1]."""
    am = np.asarray(attention_mask, dtype=np.int64)
    B, S = am.shape
    half = S // 2
    keep = am != 0
    keep[:, 0] = True
    rd = np.zeros((B, S), dtype=bool)          # READ rows: labelled positions and position 0
    rd[:, 0] = True
    if text_labels is not None:
        rd[:, :half] |= np.asarray(text_labels) != -100
    if ent_labels is not None:
        rd[:, half:] |= np.asarray(ent_labels) != -100
    keep |= rd
    keep[~(am != 0).any(axis=1)] = True
    # a sequence's packed rows: its read rows first (position order), then its other kept rows (position order)
    row_of_pos = np.full(B * S, -1, dtype=np.int32)
    pos_of_row = np.full(B * S, -1, dtype=np.int32)
    seq_offsets = np.zeros(B + 1, dtype=np.int32)
    read_offsets = np.zeros(B + 1, dtype=np.int32)
    read_rows = np.full(B * S, -1, dtype=np.int32)
    read_of_pos = np.full(B * S, -1, dtype=np.int32)
    row = nr = 0
    for b in range(B):
        first = np.nonzero(rd[b])[0]
        rest = np.nonzero(keep[b] & ~rd[b])[0]
        for j, s in enumerate(first):
            read_rows[nr + j] = row + j
            read_of_pos[b * S + s] = nr + j
        for s in np.concatenate([first, rest]):
            row_of_pos[b * S + s] = row
            pos_of_row[row] = b * S + s
            row += 1
        nr += len(first)
        seq_offsets[b + 1], read_offsets[b + 1] = row, nr
    total = row
    row_mask = np.zeros(B * S, dtype=np.int64)
    row_mask[:total] = am.reshape(-1)[pos_of_row[:total]]
    if not read:
        return row_of_pos, pos_of_row, seq_offsets, row_mask
    return row_of_pos, pos_of_row, seq_offsets, row_mask, read_rows, read_of_pos, read_offsets
