"""DrugLAMP (reference: model/DrugLAMP.py:8-78): forward(vd, vp, xd, xp, mode) -> 5-tuple / 4-tuple."""
import torch

from .basic_model import DrugLAMPBase


class DrugLAMP(DrugLAMPBase):
    two_c2p = False

    def __init__(self, n_drug_feature, n_prot_feature, n_hidden=128, **cfg):
        super().__init__(n_drug_feature, n_prot_feature, n_hidden, **cfg)

    def forward(self, vd, vp, xd, xp, mode="train"):
        vd = self.drug_extractor(vd)
        fill_p = self._fill_bit(xp)
        xd = torch.cat((xd, self._fill_bit(xd).unsqueeze(-1)), dim=-1)
        # the reference concatenates the fill bit onto the (B, 2304, 640) ESM tensor here (DrugLAMP.py:14);
        # that 641-wide copy is only consumed by the SSL head, so it is handed over as the pair
        # (embeddings, fill bit) and materialised by SSL.forward on SSL epochs only
        ssl = {"vp": vp, "xp": (xp, fill_p), "fill_bit_p": fill_p, "vd": vd, "xd": xd}
        xp = torch.cat((self._site_pool(xp), self._site_pool(fill_p.unsqueeze(-1))), dim=-1)   # (B, 256, 641)
        vpf = self._site_pool(self.protein_extractor(vp, fill_p))
        xpf, xdf = self._llm_adaptors(xp, xd)
        vpf, vdf = vpf.float(), vd.float()
        cp = {"prot": vpf, "aug_prot": xpf, "drug": vdf, "aug_drug": xdf} if self.two_c2p else None
        mv, self.A_v_gca = self._gca_branch(self.v_gca, self.v_mhla, self.v_gca_norm, vpf, vdf)
        mx, self.A_x_gca = self._gca_branch(self.x_gca, self.x_mhla, self.x_gca_norm, xpf, xdf)
        f, self.attn, self.guide_attn = self.pmma(mx, mv)
        with self._glue():
            score = self.mlp_classifier(f.mean(dim=1))
        score = score.float()
        if mode == "train":
            return vd, vpf, ssl, cp, score
        elif mode == "eval":
            return vd, vpf, score, self.attn
