"""CPU prototype for DESIGN section 8 item 7 (not on the product path): the protein CNN on DISTINCT rows only.

Two positions of a layer's output are equal whenever the token windows under their receptive fields are equal (positions
outside the sequence as a symbol of their own: a layer's zero padding is not the previous layer's output on zero tokens).  Per layer l (receptive field left_l / right_l on the token sequence) the
positions of a sample fall into classes of equal windows; one representative row per class is computed, its inputs gathered
from the previous layer's classes, BatchNorm statistics take the class sizes as weights, and the last layer is expanded back
to all positions.  Checked here against the plain computation (fp64): outputs of every position, the gradients of every
parameter and of the embedding table — autograd through gather / weighted statistics / expansion does the rest (the expansion's
backward sums the copies, a class's gradient is shared by all its members)."""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_default_dtype(torch.float64)
C, S, KS = 16, 2304, (3, 6, 9)
PADL = [(k - 1) // 2 for k in KS]                    # torch 'same': left (k - 1) // 2, right k - 1 - left
PADR = [k - 1 - l for k, l in zip(KS, PADL)]


def tiled(L, rng):
    P = L + 2
    seq = rng.integers(1, 26, size=P)
    ids = np.zeros(S, dtype=np.int64)
    for r in range(S // P):
        ids[r * P:(r + 1) * P] = seq
    return ids


def classes(ids):
    """Per layer: (class of every position, representative position of every class, class sizes)."""
    out, left, right = [], 0, 0
    # out-of-range positions get their own symbol: a layer's zero PADDING is not the previous layer's output on zero tokens
    pad = np.concatenate([np.full(32, -1, np.int64), ids, np.full(32, -1, np.int64)])
    for l in range(3):
        left, right = left + PADL[l], right + PADR[l]
        win = np.lib.stride_tricks.sliding_window_view(pad, left + right + 1)[32 - left:32 - left + S]
        _, rep, cls, cnt = np.unique(win, axis=0, return_index=True, return_inverse=True, return_counts=True)
        out.append((cls.reshape(-1), rep, cnt))
    return out


def full(ids_b, emb, ws, bs, gs, bts):
    x = emb[torch.from_numpy(ids_b)]                                               # (B, S, C); row 0 of emb is zero
    h = x.transpose(1, 2)
    for w, b, g, bt in zip(ws, bs, gs, bts):
        h = F.relu(F.conv1d(h, w, b, padding="same"))
        m = h.mean(dim=(0, 2), keepdim=True)
        v = ((h - m) ** 2).mean(dim=(0, 2), keepdim=True)
        h = (h - m) / torch.sqrt(v + 1e-5) * g.view(1, -1, 1) + bt.view(1, -1, 1)
    return h.transpose(1, 2)                                                       # (B, S, C)


def compact(ids_b, emb, ws, bs, gs, bts):
    B = ids_b.shape[0]
    cl = [classes(ids_b[b]) for b in range(B)]
    zero = torch.zeros(1, emb.shape[1])
    prev_rows = [torch.cat([emb[torch.from_numpy(ids_b[b])], zero]) for b in range(B)]  # layer-0 "classes" = positions; last row = padding
    prev_cls = [np.arange(S) for _ in range(B)]
    n_rows = 0
    for l, (w, bias, g, bt) in enumerate(zip(ws, bs, gs, bts)):
        k = KS[l]
        ys, wts = [], []
        for b in range(B):
            cls, rep, cnt = cl[b][l]
            # inputs of a class's representative: the previous layer's classes at rep - padl .. rep + padr (outside: the zero row)
            pos = rep[:, None] + np.arange(-PADL[l], PADR[l] + 1)[None, :]
            inside = (pos >= 0) & (pos < S)
            src = np.where(inside, prev_cls[b][np.clip(pos, 0, S - 1)], prev_rows[b].shape[0] - 1)
            xin = prev_rows[b][torch.from_numpy(src)]                              # (n_cls, k, C)
            y = F.relu(torch.einsum("nkc,ock->no", xin, w) + bias)                 # conv as a product over the gathered window
            ys.append(y); wts.append(torch.from_numpy(cnt).double())
        allw = torch.cat(wts); ally = torch.cat(ys)
        n = float(B * S)
        m = (ally * allw[:, None]).sum(0) / n                                      # class sizes as weights
        v = (((ally - m) ** 2) * allw[:, None]).sum(0) / n
        prev_rows = [torch.cat([(y - m) / torch.sqrt(v + 1e-5) * g + bt, zero[:, :y.shape[1]]]) for y in ys]
        prev_cls = [cl[b][l][0] for b in range(B)]
        n_rows = int(allw.numel())
    out = torch.stack([prev_rows[b][torch.from_numpy(prev_cls[b])] for b in range(B)])  # expansion: every position <- its class
    return out, n_rows


rng = np.random.default_rng(1)
Ls = [98, 398, 611, 1022, 1500]
ids_b = np.stack([tiled(L, rng) for L in Ls])
params = lambda: ([torch.randn(27, C).requires_grad_()] + [(torch.randn(C, C, k) * 0.2).requires_grad_() for k in KS] +   # noqa: E731
                  [(torch.randn(C) * 0.1).requires_grad_() for _ in range(3)] + [(1 + 0.1 * torch.randn(C)).requires_grad_() for _ in range(3)] +
                  [(0.1 * torch.randn(C)).requires_grad_() for _ in range(3)])
P0 = params()
cot = torch.randn(len(Ls), S, C)


def run(fn):
    ps = [p.detach().clone().requires_grad_() for p in P0]
    emb = ps[0] * torch.cat([torch.zeros(1, 1), torch.ones(26, 1)])                # padding_idx 0: a zero row that stays zero
    res = fn(ids_b, emb, ps[1:4], ps[4:7], ps[7:10], ps[10:13])
    out = res[0] if isinstance(res, tuple) else res
    (out * cot).sum().backward()
    return out.detach(), [p.grad for p in ps], (res[1] if isinstance(res, tuple) else None)


o_full, g_full, _ = run(full)
o_cmp, g_cmp, n_rows = run(compact)
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-300))           # noqa: E731
print("rows of the last layer: %d distinct of %d (%.1fx fewer) for L = %s" % (n_rows, len(Ls) * S, len(Ls) * S / n_rows, Ls))
print("outputs   max rel err %.1e" % rel(o_cmp, o_full))
print("gradients max rel err %.1e (embedding, 3 conv weights, 3 biases, 3 gammas, 3 betas)" % max(rel(a, b) for a, b in zip(g_cmp, g_full)))
assert rel(o_cmp, o_full) < 1e-10 and all(rel(a, b) < 1e-9 for a, b in zip(g_cmp, g_full))
print("OK")
