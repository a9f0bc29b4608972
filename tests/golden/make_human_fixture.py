"""BASELINE config 1 (DrugLAMPwoLLM on datasets/human random split, batch 32): a compact fixture of REAL rows.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_human_fixture.py

Reads /root/reference/datasets/human/random/{train,val,test}.csv (first 1024 / 256 / 256 rows) and stores DATA only:
  * per split: (drug index, protein index, label Y) triples — unique SMILES / unique protein sequences are numbered in
    order of first appearance (what the reference's meta dict calls Drug_ID / Prot_ID, handler/dataset.py:197-205);
  * per unique protein: its residue codes as the REFERENCE's own `repeat_integer_label_protein` (utils.py:392-412)
    produces them (one period: codes of sequence[:1022]); the tiling to 2304 positions is restated in
    druglamp_amd/data.py and pinned by four full 2304-long encodings stored here;
  * per unique drug: the number of atom symbols in the SMILES string (a stand-in for the node count; rdkit / dgllife
    are absent, so graph FEATURES stay synthetic — SURVEY 8d "C1: real CSV rows for ids / labels, synthetic features").
No reference source text is stored."""
import csv
import os
import re
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

import ref_harness  # noqa: E402

ref_harness.import_reference()
import utils as RU  # noqa: E402  (the reference's utils.py, through the harness stubs)

ROOT = "/root/reference/datasets/human/random"
LIMIT = {"train": 1024, "val": 256, "test": 256}
ATOM = re.compile(r"Cl|Br|Si|Se|Na|Li|Mg|Zn|Fe|Cu|Ca|Al|Ag|Au|Pt|Hg|[BCNOFPSIKHcnosp]")

drugs, prots, rows = {}, {}, {}
for split, lim in LIMIT.items():
    out = []
    with open(os.path.join(ROOT, split + ".csv")) as f:
        for i, r in enumerate(csv.DictReader(f)):
            if i >= lim:
                break
            d = drugs.setdefault(r["SMILES"], len(drugs))
            p = prots.setdefault(r["Protein"], len(prots))
            out.append((d, p, int(float(r["Y"]))))
    rows[split] = np.asarray(out, dtype=np.int32)

codes, offs = [], [0]
full = {}
for seq, pid in prots.items():
    enc = RU.repeat_integer_label_protein(seq, 1022)            # the reference's own encoder, (2304,) float64
    L = min(len(seq), 1022)
    codes.append(enc[1:1 + L].astype(np.uint8))                  # one period (position 0 is the CLS slot = 0)
    offs.append(offs[-1] + L)
    if pid < 4:
        full[pid] = enc.astype(np.uint8)
n_atoms = np.zeros(len(drugs), dtype=np.int16)
for smi, did in drugs.items():
    n_atoms[did] = max(1, min(512, len(ATOM.findall(smi))))
np.savez_compressed(os.path.join(HERE, "human_random_rows.npz"), train=rows["train"], val=rows["val"], test=rows["test"],
                    prot_codes=np.concatenate(codes), prot_offsets=np.asarray(offs, dtype=np.int64), drug_atoms=n_atoms,
                    full_encodings=np.stack([full[i] for i in range(4)]))
print("rows", {k: v.shape for k, v in rows.items()}, "unique drugs", len(drugs), "unique proteins", len(prots),
      "size", os.path.getsize(os.path.join(HERE, "human_random_rows.npz")))
