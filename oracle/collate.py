"""CPU restatement of the reference's LLM-embedding collate padding (TEST INFRASTRUCTURE ONLY, like the rest of
oracle/: imported by tests/, never by druglamp_amd/).

  tail_pad   (reference utils.py:304-312): out[i, :len_i] = x_i, zeros after; used for the drug embeddings (maxsize 512).
  repeat_pad (reference utils.py:314-324): x_i is written floor(maxsize / len_i) times back to back, zeros after
             (so a sequence longer than maxsize yields an all-zero row block); used for the protein embeddings
             (maxsize 9 * 256), reference utils.py:326-334.
Pinned by tests/golden/collate_pad.npz (outputs of the reference's own functions on detgen inputs)."""
import numpy as np


def tail_pad(xs, maxsize):
    out = np.zeros((len(xs), maxsize, xs[0].shape[-1]), dtype=np.float32)
    for i, a in enumerate(xs):
        out[i, :a.shape[-2], :] = a
    return out


def repeat_pad(xs, maxsize):
    out = np.zeros((len(xs), maxsize, xs[0].shape[-1]), dtype=np.float32)
    for i, a in enumerate(xs):
        n = a.shape[-2]
        for j in range(maxsize // n):
            out[i, j * n:(j + 1) * n, :] = a
    return out


def ragged_inputs(tag, lengths, feat):
    """Deterministic ragged inputs shared by the golden generator and the tests."""
    from . import detgen
    return [detgen.normalish("%s.%d" % (tag, i), (n, feat)) for i, n in enumerate(lengths)]
