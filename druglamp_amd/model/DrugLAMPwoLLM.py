"""DrugLAMPwoLLM (reference: model/DrugLAMPwoLLM.py:8-51): no LLM branch; pmma(mv, mv)."""
from .. import functional as Fn
from .. import ops
from .basic_model import DrugLAMPBase


class DrugLAMPwoLLM(DrugLAMPBase):
    def __init__(self, n_drug_feature, n_prot_feature, n_hidden=128, **cfg):
        super().__init__(n_drug_feature, n_prot_feature, n_hidden, **cfg)

    def forward(self, vd, vp, xd, xp, mode="train", hints=None):
        """hints (protein_plan.BatchHints, optional): host-side knowledge about the batch's padding structure — the compact
        forms it enables are verified on the device.  Without it every row is computed."""
        with Fn.deferred_bn_ticks():             # the nine BatchNorm step counters advance in one launch at the end
            return self._forward(vd, vp, xd, xp, mode, hints)

    def _forward(self, vd, vp, xd, xp, mode, hints=None):
        vd = self.drug_extractor(vd)
        fill_p, _ = ops.fill_pool(xp, self.site_len, self.compute_dtype)
        ssl = {"vp": vp, "xp": None, "fill_bit_p": fill_p, "vd": vd, "xd": None, "p_mode": "vp"}
        vpc = self.protein_extractor(vp, fill_p, site_pool=self.site_len, plan=self._protein_plan(hints, vp))       # compute dtype
        vpf = vpc.float()
        mv, self.A_v_gca = self._gca_branch(self.v_gca, self.v_mhla, self.v_gca_norm, vpc, Fn.cast(vd, self.compute_dtype), raw=(hints is None or hints.raw_attention),
                                            tail=getattr(vd, "_dl_tail", None))
        f, self.attn, self.guide_attn = self.pmma(mv, mv)
        with self._glue():
            score = self.mlp_classifier(Fn.TokenMeanFn.apply(f))
        score = score.float()
        if mode == "train":
            return vd, vpf, ssl, None, score
        elif mode == "eval":
            return vd, vpf, score, self.attn
