from .VIPRS import VIPRS
from .VIPRSMix import VIPRSMix

__all__ = ["VIPRS", "VIPRSMix"]
