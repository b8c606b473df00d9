"""Reads the reference's ``config.toml`` unchanged (same keys and types as
src/data/config.py:8-85; the three directory fields become ``Path``)."""

from __future__ import annotations

from pathlib import Path

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # the build image runs 3.10
    import tomli as _toml

_REQUIRED = {
    "training": ("batch_size", "random_seed", "training_steps", "image_buffer_size", "style_mixing_prob",
                 "deterministic_cuda_kernels", "gpu_number", "checkpoint_directory", "training_run"),
    "optimisation": ("style_cycle_loss_lambda", "identity_loss_lambda", "reconstruction_loss_lambda",
                     "kl_loss_lambda", "path_loss_lambda", "path_loss_jacobian_granularity", "learning_rate",
                     "mapping_network_learning_rate", "adam_betas"),
    "ada": ("discriminator_real_acc_target", "ada_overfitting_measurement_n_images", "ada_adjustment_size"),
    "evaluation": ("log_interval", "checkpoint_interval", "n_evaluation_images", "inference_batch_size"),
    "architecture": ("w_dim", "add_latent_noise", "min_latent_resolution", "n_resnet_blocks",
                     "mapping_network_layers"),
    "data": ("image_size", "image_channels", "shoemark_data_dir", "shoeprint_data_dir"),
}


def load_config(path) -> dict:
    """Load a TOML file of hyper-parameters into the nested dict the step functions take."""
    with Path(path).open("rb") as f:
        config = _toml.load(f)
    for section, keys in _REQUIRED.items():
        if section not in config:
            raise KeyError(f"config section [{section}] is missing")
        for k in keys:
            if k not in config[section]:
                raise KeyError(f"config key {section}.{k} is missing")
    config["training"]["checkpoint_directory"] = Path(config["training"]["checkpoint_directory"])
    config["data"]["shoeprint_data_dir"] = Path(config["data"]["shoeprint_data_dir"])
    config["data"]["shoemark_data_dir"] = Path(config["data"]["shoemark_data_dir"])
    return config
