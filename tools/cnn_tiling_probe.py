"""CPU probe for DESIGN section 8 item 7: the reference tiles a protein of L residues (L + 2 tokens with the two specials) to
2304 positions and zero-fills the rest.  Three Conv1d('same', k = 3 / 6 / 9) + ReLU + BatchNorm layers (BatchNorm: per-channel
affine, position-wise) see a periodic input, so their output is periodic with period L + 2 except near the seams.  Prints,
for a few L, how many of the 2304 output positions are DISTINCT rows and the measured receptive field."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
C, S = 128, 2304
emb = torch.randn(27, C)
ws = [torch.randn(C, C, k) * 0.05 for k in (3, 6, 9)]
bs = [torch.randn(C) * 0.1 for _ in range(3)]
for L in (98, 398, 1022):
    P = L + 2
    seq = torch.randint(1, 26, (P,))
    reps = S // P
    ids = torch.zeros(S, dtype=torch.long)
    for r in range(reps):
        ids[r * P:(r + 1) * P] = seq
    x = emb[ids].t().unsqueeze(0)                      # (1, C, S); id 0 = padding token
    x[:, :, ids == 0] = 0
    h = x
    for w, b in zip(ws, bs):
        h = F.relu(F.conv1d(h, w, b, padding="same"))  # BatchNorm omitted: a per-channel affine map does not change which rows are equal
    rows = h[0].t()                                    # (S, C)
    # interior periodicity: position t equals t + P wherever both receptive fields lie inside the tiled region
    eq = (rows[:S - P] == rows[P:]).all(dim=1)
    first_ok = int(eq.nonzero()[0]) if eq.any() else -1
    last_ok = int(eq[:reps * P - P].nonzero()[-1]) if eq[:reps * P - P].any() else -1
    uniq = torch.unique(rows, dim=0).shape[0]
    print("L = %4d: period %4d x %d repetitions + %3d zero positions -> %4d distinct rows of %d (%.1fx fewer); t == t + P from t = %d to t = %d "
          "(left field %d, right field %d)" % (L, P, reps, S - reps * P, uniq, S, S / uniq, first_ok, last_ok, first_ok, reps * P - P - 1 - last_ok))
