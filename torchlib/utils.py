"""`torchlib.utils` names of the hot path, re-exported from primia_amd (see torchlib/__init__.py)."""
from primia_amd.datapipe import MixUp, To_one_hot, calc_class_weights, calc_mean_std  # noqa: F401
from primia_amd.syft_compat import setup_pysyft  # noqa: F401
from primia_amd.torchlib_compat import (  # noqa: F401
    Arguments,
    Cross_entropy_one_hot,
    LearningRateScheduler,
    aggregation,
    save_config_results,
    save_model,
    secure_aggregation_epoch,
    send_new_models,
    stats_table,
    test,
    train,
    train_federated,
)
