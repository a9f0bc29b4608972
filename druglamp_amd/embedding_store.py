"""Device-resident store of the pre-extracted LLM embeddings.

The reference keeps one `.pt` file per unique drug / protein, `torch.load`s two of them per sample
(handler/dataset.py:186-195) and pads them on the host in Python loops (utils.py:304-334).  An MI355X has 288 GB of
HBM: every unique entity's (len, F) embedding is packed ONCE into one contiguous device tensor, and a training batch
is assembled by one `dl_gather_pad` launch per modality straight into the (B, S, F) layout the model consumes
(repeat-padding for proteins, tail-padding for drugs) — no per-sample I/O, no host loops, no H2D on the step."""
from __future__ import annotations

from typing import Dict, Hashable, Iterable, Sequence, Tuple

import numpy as np
import torch

from . import ops


class EmbeddingStore:
    def __init__(self, feat_dim: int, dtype: torch.dtype = torch.bfloat16, device="cuda"):
        self.feat_dim, self.dtype, self.device = feat_dim, dtype, torch.device(device)
        self._index: Dict[Hashable, Tuple[int, int]] = {}
        self._chunks, self._rows = [], 0
        self._store = None
        self._finalized = False

    def add(self, key: Hashable, emb) -> None:
        """emb: (len, feat_dim) array / tensor.  Keys are unique entities (Drug_ID / Prot_ID)."""
        if key in self._index:
            return
        t = torch.as_tensor(np.asarray(emb) if not torch.is_tensor(emb) else emb)
        if t.dim() != 2 or t.shape[1] != self.feat_dim:
            raise ValueError("EmbeddingStore.add: expected (len, %d), got %s" % (self.feat_dim, tuple(t.shape)))
        self._index[key] = (self._rows, t.shape[0])
        # chunks appended after a finalize() join the device-resident store on the device (torch.cat needs one device)
        self._chunks.append(t.to(device=self.device if self._finalized else t.device, dtype=self.dtype))
        self._rows += t.shape[0]
        self._store = None

    def finalize(self) -> "EmbeddingStore":
        self._store = torch.cat([c.to(self.device) for c in self._chunks], dim=0).contiguous()
        self._chunks = [self._store]            # keep one reference so that add() after finalize() still works
        self._finalized = True
        return self

    # ---- on-disk format (replaces the reference's one `.pt` file per entity, handler/dataset.py:119-122,186-195) ----
    # One file per modality, little endian:
    #   [0:8)    magic b"DLEMBST1"
    #   [8:40)   uint32 dtype code (0 = float32, 1 = bfloat16), uint32 feat_dim, uint64 n_entities, uint64 total_rows, uint64 key_bytes
    #   then     int64 row_offset[n_entities], int32 length[n_entities], the keys as one UTF-8 JSON list (key_bytes), zero padding
    #            to a 4096-byte boundary, and the rows themselves: total_rows x feat_dim elements, entity after entity.
    # The row block is read (or memory-mapped) as ONE tensor and goes to the device with a single copy.
    MAGIC = b"DLEMBST1"

    def save(self, path: str) -> None:
        import json
        import struct
        if self._store is None or self._store.shape[0] != self._rows:
            self.finalize()
        keys = list(self._index.keys())
        if not all(isinstance(k, (int, str)) for k in keys):
            raise TypeError("EmbeddingStore.save: keys must be int or str")
        kb = json.dumps(keys).encode("utf-8")
        offs = np.asarray([self._index[k][0] for k in keys], dtype="<i8")
        lens = np.asarray([self._index[k][1] for k in keys], dtype="<i4")
        code = {torch.float32: 0, torch.bfloat16: 1}[self.dtype]
        head = self.MAGIC + struct.pack("<IIQQQ", code, self.feat_dim, len(keys), self._rows, len(kb))
        body = head + offs.tobytes() + lens.tobytes() + kb
        pad = (-len(body)) % 4096
        rows = self._store.detach().cpu().contiguous()
        raw = rows.view(torch.int16).numpy() if self.dtype == torch.bfloat16 else rows.numpy()
        with open(path, "wb") as f:
            f.write(body + b"\0" * pad)
            f.write(raw.tobytes())

    @classmethod
    def load(cls, path: str, device="cuda", mmap: bool = True) -> "EmbeddingStore":
        import json
        import struct
        with open(path, "rb") as f:
            head = f.read(40)
            if head[:8] != cls.MAGIC:
                raise ValueError("EmbeddingStore.load: %s is not an embedding store file" % path)
            code, feat, n, total, klen = struct.unpack("<IIQQQ", head[8:40])
            offs = np.frombuffer(f.read(8 * n), dtype="<i8")
            lens = np.frombuffer(f.read(4 * n), dtype="<i4")
            keys = json.loads(f.read(klen).decode("utf-8"))
            start = 40 + 12 * n + klen
            start += (-start) % 4096
        dtype = {0: torch.float32, 1: torch.bfloat16}[code]
        npdt = np.float32 if code == 0 else np.int16
        arr = np.memmap(path, dtype=npdt, mode="r", offset=start, shape=(total, feat)) if mmap else \
            np.fromfile(path, dtype=npdt, offset=start).reshape(total, feat)
        st = cls(feat, dtype=dtype, device=device)
        st._index = {k: (int(o), int(m)) for k, o, m in zip(keys, offs, lens)}
        st._rows = int(total)
        # the row block goes to the device in slices of <= 256 MB: a memory-mapped file of hundreds of GB never has to be
        # resident in host memory as a whole
        dst = torch.empty((total, feat), dtype=dtype, device=st.device)
        rows_per = max(1, (256 << 20) // max(1, feat * arr.dtype.itemsize))
        for r0 in range(0, int(total), rows_per):
            piece = torch.from_numpy(np.ascontiguousarray(arr[r0:r0 + rows_per]))
            dst[r0:r0 + rows_per].copy_(piece.view(torch.bfloat16) if code == 1 else piece)
        st._store = dst
        st._chunks = [st._store]
        st._finalized = True
        return st

    @property
    def nbytes(self) -> int:
        return self._rows * self.feat_dim * torch.empty((), dtype=self.dtype).element_size()

    def length(self, key: Hashable) -> int:
        """Rows stored under `key` (the collate's token count: what Trainer turns into a padding hint)."""
        return int(self._index[key][1])

    def batch(self, keys: Sequence[Hashable], max_rows: int, repeat: bool) -> torch.Tensor:
        """(B, max_rows, feat_dim): repeat=True is the reference's repeat_pad, repeat=False its tail_pad."""
        if self._store is None or self._store.shape[0] != self._rows:
            self.finalize()
        idx = [self._index[k] for k in keys]
        if not repeat:
            # the reference's tail_pad (utils.py:304-312) raises on a sequence longer than maxsize (shape mismatch in
            # `out[i, :len] = a`); its repeat_pad yields an all-zero block instead (0 whole repetitions), kept as is
            too_long = [k for k, (_, n) in zip(keys, idx) if n > max_rows]
            if too_long:
                raise ValueError("EmbeddingStore.batch: %d sequence(s) longer than max_rows=%d under tail padding (first key: %r)"
                                 % (len(too_long), max_rows, too_long[0]))
        offsets = torch.tensor([o for o, _ in idx], dtype=torch.int64, device=self.device)
        lengths = torch.tensor([n for _, n in idx], dtype=torch.int32, device=self.device)
        return ops.gather_pad(self._store, offsets, lengths, max_rows, repeat)


def collate_llm(prot_store: EmbeddingStore, drug_store: EmbeddingStore, prot_keys: Iterable[Hashable],
                drug_keys: Iterable[Hashable], prot_rows: int = 9 * 256, drug_rows: int = 512):
    """(d_llm (B, 512, Dd), p_llm (B, 2304, Dp)) as multimodality_collate_func yields them (utils.py:326-334)."""
    return drug_store.batch(list(drug_keys), drug_rows, repeat=False), prot_store.batch(list(prot_keys), prot_rows, repeat=True)
