"""Golden vectors for the collate padding (reference utils.py:304-324), produced by the REAL reference functions.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_collate_golden.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_harness  # noqa: E402
from oracle import collate  # noqa: E402

ref_harness.import_reference()
import utils as ref_utils  # noqa: E402  (the reference's utils.py)

MAXSIZE, FEAT = 48, 8
TAIL_LENS = [1, 5, 7, 16, 47, 48]
REP_LENS = [1, 5, 7, 16, 24, 25, 47, 48, 49, 60]
tail_in = collate.ragged_inputs("collate.tail", TAIL_LENS, FEAT)
rep_in = collate.ragged_inputs("collate.rep", REP_LENS, FEAT)
tail_out = ref_utils.tail_pad([torch.from_numpy(a) for a in tail_in], MAXSIZE).numpy()
rep_out = ref_utils.repeat_pad([torch.from_numpy(a) for a in rep_in], MAXSIZE).numpy()
np.savez_compressed(os.path.join(HERE, "collate_pad.npz"), maxsize=MAXSIZE, feat=FEAT, tail_lens=np.array(TAIL_LENS),
                    rep_lens=np.array(REP_LENS), tail_out=tail_out, rep_out=rep_out)
print("wrote collate_pad.npz", tail_out.shape, rep_out.shape)
