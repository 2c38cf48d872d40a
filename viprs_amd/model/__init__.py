from .VIPRS import VIPRS
from .VIPRSMix import VIPRSMix
from .VIPRSPerChromosome import VIPRSPerChromosome
from .VIPRSMixPerChromosome import VIPRSMixPerChromosome
from .gridsearch import HyperparameterGrid, VIPRSGrid, bayesian_model_average, select_best_model

__all__ = ["VIPRS", "VIPRSMix", "VIPRSPerChromosome", "VIPRSMixPerChromosome", "VIPRSGrid", "HyperparameterGrid", "select_best_model", "bayesian_model_average"]
