"""Drop-in import surface of the reference (`from Algorithms import PCA_BO`, reference:
Algorithms/__init__.py:1-3), backed by the MI355X HIP library instead of botorch/gpytorch/sklearn.
Put this directory's parent (`para-ortho-pca-bo_amd/`) on sys.path."""
from .BayesianOptimization.Vanilla_BO import Vanilla_BO  # noqa: F401
from .BayesianOptimization.PCA_BO import PCA_BO  # noqa: F401
from .BayesianOptimization.AbstractBayesianOptimizer import AbstractBayesianOptimizer, LHS_sampler  # noqa: F401
from .AbstractAlgorithm import AbstractAlgorithm  # noqa: F401
from .Experiment.ExperimentRunner import ExperimentRunner  # noqa: F401
