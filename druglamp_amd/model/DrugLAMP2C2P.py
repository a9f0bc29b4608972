"""DrugLAMP2C2P (reference: model/DrugLAMP2C2P.py:8-89): DrugLAMP plus the cross-modality input dict."""
from .DrugLAMP import DrugLAMP


class DrugLAMP2C2P(DrugLAMP):
    two_c2p = True
