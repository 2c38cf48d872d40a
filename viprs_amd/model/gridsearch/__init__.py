from .HyperparameterGrid import HyperparameterGrid
from .VIPRSGrid import VIPRSGrid
from .grid_utils import bayesian_model_average, select_best_model

__all__ = ["HyperparameterGrid", "VIPRSGrid", "select_best_model", "bayesian_model_average"]
