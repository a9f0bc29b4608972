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
        xpc, xdc = self._llm_adaptors(xp, xd, hints.drug_tokens if hints is not None else 0)
        vpf = vpc.float()                                                       # fp32 copies only for the returned tuple
        cp = {"prot": vpf, "aug_prot": xpc.float(), "drug": vd.float(), "aug_drug": xdc.float()} if self.two_c2p else None
        mv, self.A_v_gca = self._gca_branch(self.v_gca, self.v_mhla, self.v_gca_norm, vpc, vdc)
        mx, self.A_x_gca = self._gca_branch(self.x_gca, self.x_mhla, self.x_gca_norm, xpc, xdc)
        f, self.attn, self.guide_attn = self.pmma(mx, mv)
        with self._glue():
            score = self.mlp_classifier(Fn.TokenMeanFn.apply(f))
        score = score.float()
        if mode == "train":
            return vd, vpf, ssl, cp, score
        elif mode == "eval":
            return vd, vpf, score, self.attn
