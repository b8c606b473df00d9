"""Reads the reference's ``config.toml`` unchanged: same sections, keys and value types as the
typed dictionaries of src/data/config.py:8-65, the three directory entries turned into ``Path``
objects (src/data/config.py:76-82).  Unlike the reference this loader VALIDATES: a missing section
or key raises ``KeyError`` naming it, instead of surfacing later as a lookup error mid-training."""

from __future__ import annotations

from pathlib import Path

try:  # Python >= 3.11
    import tomllib as _toml
except ModuleNotFoundError:  # the build image runs 3.10
    import tomli as _toml

# section -> keys the step functions, train.py and the checkpoint code read
_SCHEMA = {
    section: tuple(keys.split())
    for section, keys in (
        ("training", "batch_size random_seed training_steps image_buffer_size style_mixing_prob "
                     "deterministic_cuda_kernels gpu_number checkpoint_directory training_run"),
        ("optimisation", "style_cycle_loss_lambda identity_loss_lambda reconstruction_loss_lambda kl_loss_lambda "
                         "path_loss_lambda path_loss_jacobian_granularity learning_rate "
                         "mapping_network_learning_rate adam_betas"),
        ("ada", "discriminator_real_acc_target ada_overfitting_measurement_n_images ada_adjustment_size"),
        ("evaluation", "log_interval checkpoint_interval n_evaluation_images inference_batch_size"),
        ("architecture", "w_dim add_latent_noise min_latent_resolution n_resnet_blocks mapping_network_layers"),
        ("data", "image_size image_channels shoemark_data_dir shoeprint_data_dir"),
    )
}
_PATH_FIELDS = (("training", "checkpoint_directory"), ("data", "shoeprint_data_dir"), ("data", "shoemark_data_dir"))


def load_config(path) -> dict:
    """TOML file of hyper-parameters -> the nested dict every step function takes."""
    with Path(path).open("rb") as handle:
        cfg = _toml.load(handle)
    for section, keys in _SCHEMA.items():
        if section not in cfg:
            raise KeyError(f"config section [{section}] is missing")
        absent = [k for k in keys if k not in cfg[section]]
        if absent:
            raise KeyError(f"config key {section}.{absent[0]} is missing")
    for section, key in _PATH_FIELDS:
        cfg[section][key] = Path(cfg[section][key])
    return cfg
