"""DrugLAMP (reference: model/DrugLAMP.py:8-78): forward(vd, vp, xd, xp, mode) -> 5-tuple / 4-tuple."""
import torch

from .. import functional as Fn
from .. import ops
from .basic_model import DrugLAMPBase


class DrugLAMP(DrugLAMPBase):
    two_c2p = False

    def __init__(self, n_drug_feature, n_prot_feature, n_hidden=128, **cfg):
        super().__init__(n_drug_feature, n_prot_feature, n_hidden, **cfg)

    def forward(self, vd, vp, xd, xp, mode="train", hints=None):
        """hints (protein_plan.BatchHints, optional): host-side knowledge about the batch's padding structure — the compact
        forms it enables are verified on the device.  Without it every row is computed."""
        with Fn.deferred_bn_ticks():             # the nine BatchNorm step counters advance in one launch at the end
            return self._forward(vd, vp, xd, xp, mode, hints)

    def _forward(self, vd, vp, xd, xp, mode, hints=None):
        if self.branch_streams and hints is not None and hints.branch_streams and xp.is_cuda:
            return self._forward_branches(vd, vp, xd, xp, mode, hints)
        vd = self.drug_extractor(vd)
        # one pass over each LLM tensor: fill bit + (site-pooled) fill-bit-augmented features, already padded
        # to the GEMM alignment (641 -> 648, 385 -> 392 columns)
        cdt = self.compute_dtype
        fill_p, xps = ops.fill_pool(xp, self.site_len, cdt)
        fill_d, xdp = ops.fill_pool(xd, 1, cdt)
        # the reference concatenates the fill bits onto the raw tensors here (DrugLAMP.py:11-19); those copies
        # are only consumed by the SSL head, so they are handed over as (tensor, fill bit) pairs and
        # materialised by SSL.forward on SSL epochs only
        ssl = {"vp": vp, "xp": (xp, fill_p), "fill_bit_p": fill_p, "vd": vd, "xd": (xdp, xd.shape[-1] + 1)}   # drug LLM features + fill bit, zero-padded to 392 columns
        xp, xd = xps, xdp
        vpc = self.protein_extractor(vp, fill_p, site_pool=self.site_len, plan=self._protein_plan(hints, vp))       # compute dtype
        vdc = Fn.cast(vd, cdt)
        vtail = getattr(vd, "_dl_tail", None)                                   # (distinct rows, multiplicity) of the compact forms
        xpc, xdc = self._llm_adaptors(xp, xd, hints.drug_tokens if hints is not None else 0)
        xtail = getattr(xdc, "_dl_tail", None)
        ssl["drug_rows"] = self._ssl_drug_rows(vtail, xtail)
        vpf = vpc.float()                                                       # fp32 copies only for the returned tuple
        cp = {"prot": vpf, "aug_prot": xpc.float(), "drug": vd.float(), "aug_drug": xdc.float()} if self.two_c2p else None
        mv, self.A_v_gca = self._gca_branch(self.v_gca, self.v_mhla, self.v_gca_norm, vpc, vdc, raw=(hints is None or hints.raw_attention), tail=vtail)
        mx, self.A_x_gca = self._gca_branch(self.x_gca, self.x_mhla, self.x_gca_norm, xpc, xdc, raw=(hints is None or hints.raw_attention), tail=xtail)
        f, self.attn, self.guide_attn = self.pmma(mx, mv)
        with self._glue():
            score = self.mlp_classifier(Fn.TokenMeanFn.apply(f))
        score = score.float()
        if mode == "train":
            return vd, vpf, ssl, cp, score
        elif mode == "eval":
            return vd, vpf, score, self.attn

    # ---- the same forward with its independent branches on side HIP streams (default; DL_BRANCH_STREAMS=0: one stream) ------
    # MolecularGCN, ProteinCNN and the two LLM adaptors do not depend on each other; neither do the v / x cross-attention
    # branches (PGCA -> MHLA -> LayerNorm).  Each is a chain of kernels that leave part of the chip idle (few-workgroup
    # launches at the strong-scaling batches; epilogue / reduction tails at batch 256): on separate streams they share it.
    # Autograd runs a node's backward on the stream of its forward, so the backward passes of the branches overlap the same
    # way; a captured step records them as parallel branches of the hipGraph.  Arithmetic is unchanged (every kernel is
    # deterministic; scratch buffers are per stream; the weight-image refresh runs before the fork).
    def _forward_branches(self, vd, vp, xd, xp, mode, hints=None):
        cdt = self.compute_dtype
        cur = torch.cuda.current_stream()
        Fn.ensure_images_current()               # the one-launch weight-image refresh, before any branch reads an image
        if self._streams is None:
            self._streams = tuple(torch.cuda.Stream() for _ in range(3))
        sa, sb, sc = self._streams
        sa.wait_stream(cur)
        with torch.cuda.stream(sa):              # branch a: drug graph -> MolecularGCN
            vd = self.drug_extractor(vd)
            vdc = Fn.cast(vd, cdt)
            vtail = getattr(vd, "_dl_tail", None)
        fill_p, xps = ops.fill_pool(xp, self.site_len, cdt)
        sb.wait_stream(cur)
        with torch.cuda.stream(sb):              # branch b: protein LLM adaptor
            xpc = self._prot_adaptor(xps)
        fill_d, xdp = ops.fill_pool(xd, 1, cdt)
        sc.wait_stream(cur)
        with torch.cuda.stream(sc):              # branch c: drug LLM adaptor
            xdc = self._drug_adaptor(xdp, hints.drug_tokens if hints is not None else 0)
            xtail = getattr(xdc, "_dl_tail", None)
        ssl = {"vp": vp, "xp": (xp, fill_p), "fill_bit_p": fill_p, "vd": vd, "xd": (xdp, xd.shape[-1] + 1),
               "drug_rows": self._ssl_drug_rows(vtail, xtail)}
        vpc = self.protein_extractor(vp, fill_p, site_pool=self.site_len, plan=self._protein_plan(hints, vp))   # main stream
        # v branch (needs the CNN and the GCN) continues on stream a; the x branch (needs both adaptors) runs on the main
        # stream behind the CNN — side streams only ever fork from and join the main stream (a side-to-side dependency made
        # hipStreamEndCapture of the step crash on ROCm 7.2)
        sa.wait_stream(cur)
        with torch.cuda.stream(sa):
            mv, self.A_v_gca = self._gca_branch(self.v_gca, self.v_mhla, self.v_gca_norm, vpc, vdc, raw=(hints is None or hints.raw_attention), tail=vtail)
        cur.wait_stream(sb)
        cur.wait_stream(sc)
        mx, self.A_x_gca = self._gca_branch(self.x_gca, self.x_mhla, self.x_gca_norm, xpc, xdc, raw=(hints is None or hints.raw_attention), tail=xtail)
        cur.wait_stream(sa)
        # produced on one stream, consumed (or freed) on another: tell the caching allocator
        for t, sts in ((vd, (cur,)), (xpc, (cur,)), (xdc, (cur,)), (mv, (cur,)), (xps, (sb,)), (xdp, (sc,)), (vpc, (sa,))):
            for st in sts:
                t.record_stream(st)
        if xtail is not None:
            xtail[0].record_stream(cur)
        vpf = vpc.float()
        cp = {"prot": vpf, "aug_prot": xpc.float(), "drug": vd.float(), "aug_drug": xdc.float()} if self.two_c2p else None
        f, self.attn, self.guide_attn = self.pmma(mx, mv)
        with self._glue():
            score = self.mlp_classifier(Fn.TokenMeanFn.apply(f))
        score = score.float()
        if mode == "train":
            return vd, vpf, ssl, cp, score
        elif mode == "eval":
            return vd, vpf, score, self.attn
