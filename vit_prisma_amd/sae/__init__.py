"""SAE subsystem behind the reference's API: config, module, activation store, trainer."""
from .config import VisionModelSAERunnerConfig
from .sae import SparseAutoencoder, StandardSparseAutoencoder, TopK, get_activation_fn
from .store import CacheVisionActivationStore, VisionActivationsStore
from .trainer import VisionSAETrainer
from .variants import GatedSparseAutoencoder, Transcoder

__all__ = ["CacheVisionActivationStore", "VisionModelSAERunnerConfig", "SparseAutoencoder", "StandardSparseAutoencoder", "TopK",
           "get_activation_fn", "VisionActivationsStore", "VisionSAETrainer", "GatedSparseAutoencoder", "Transcoder"]
