"""SAE subsystem behind the reference's API: config, module, activation store, trainer."""
from .config import VisionModelSAERunnerConfig
from .sae import SparseAutoencoder, StandardSparseAutoencoder, TopK, get_activation_fn
from .store import VisionActivationsStore
from .trainer import VisionSAETrainer

__all__ = ["VisionModelSAERunnerConfig", "SparseAutoencoder", "StandardSparseAutoencoder", "TopK",
           "get_activation_fn", "VisionActivationsStore", "VisionSAETrainer"]
