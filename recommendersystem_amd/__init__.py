"""MI355X-native training hot path of Fro116/RecommenderSystem.

Only what the path needs: csrc/ (HIP kernels + C ABI, built into librsys_hip.so)
and the host-side mirror of the reference's model / optimizer / training-loop API.
"""
from ._lib import RsysError, device_count, lib  # noqa: F401
from .model import ALL_MEDIUMS, ALL_METRICS, RecommenderModel, synchronize  # noqa: F401
from .optim import create_optimizer  # noqa: F401
from .train import (ConstantScheduler, EarlyStopper, WSDScheduler, evaluate_metrics, make_early_stopper,  # noqa: F401
                    make_task_weights, minimize_quadratic, train_epoch, wsum)
from . import checkpoint, cli, data, dist, h5, serve, shards, train, workload  # noqa: F401,E402  (ra.data.FinetuneDataset, ra.train.train, ...)
