"""Counterpart of the reference's src/trainer.py: same hard-coded hyper-parameter surface, same call sequence
(load_indexes -> PatchHandler3D.initialize_dataset -> TrainerController -> init_model_dir -> train_network).
Launch with `python -m torch.distributed.run --nproc-per-node N scripts/trainer.py` for data-parallel training.

The input pipeline is the measured one: DevicePatchHandler3D keeps the decoded HDF5 volumes in HBM and cuts / rotates / normalises
every batch on the device (bit-identical to the host loader, tests/test_gpu_device_loader.py).  FDN_HOST_LOADER=1 selects the host
loader instead -- PatchHandler3D with its prefetch thread, worker pool and pinned staging buffers -- e.g. when the dataset does
not fit beside the activations."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
data = importlib.import_module("4dflownet_amd.data")
parallel = importlib.import_module("4dflownet_amd.parallel")
trainer = importlib.import_module("4dflownet_amd.trainer")


def make_handler(data_dir, patch_size, res_increase, batch_size, mask_threshold):
    """(handler, kwargs for initialize_dataset): the on-device loader unless FDN_HOST_LOADER is set."""
    if os.environ.get("FDN_HOST_LOADER", "0") not in ("", "0"):
        return data.PatchHandler3D(data_dir, patch_size, res_increase, batch_size, mask_threshold), {"pinned": True}
    data_device = importlib.import_module("4dflownet_amd.data_device")
    return data_device.DevicePatchHandler3D(data_dir, patch_size, res_increase, batch_size, mask_threshold), {}


if __name__ == "__main__":
    data_dir = os.environ.get("FDN_DATA_DIR", '../data')
    training_file = '{}/train.csv'.format(data_dir)
    validate_file = '{}/validate.csv'.format(data_dir)
    QUICKSAVE = True
    benchmark_file = '{}/benchmark.csv'.format(data_dir)
    restore = False
    if restore:
        model_dir = "../models/4DFlowNet"
        model_file = "4DFlowNet-best.h5"

    # Hyperparameters optimisation variables (trainer.py:28-39)
    initial_learning_rate = 2e-4
    epochs = 60
    batch_size = 20
    mask_threshold = 0.6
    network_name = '4DFlowNet'
    patch_size = 16
    res_increase = 2
    low_resblock = 8
    hi_resblock = 4

    parallel.init_from_env()
    trainset = data.load_indexes(training_file)
    valset = data.load_indexes(validate_file)
    z, kw = make_handler(data_dir, patch_size, res_increase, batch_size, mask_threshold)
    trainset = z.initialize_dataset(trainset, shuffle=True, n_parallel=None, **kw)
    valdh, kw = make_handler(data_dir, patch_size, res_increase, batch_size, mask_threshold)
    valset = valdh.initialize_dataset(valset, shuffle=True, n_parallel=None, **kw)
    testset = None
    if QUICKSAVE and benchmark_file is not None:
        benchmark_set = data.load_indexes(benchmark_file)
        ph, kw = make_handler(data_dir, patch_size, res_increase, batch_size, mask_threshold)
        testset = ph.initialize_dataset(benchmark_set, shuffle=False, shard=(0, 1), **kw)

    print("4DFlowNet Patch %d, lr %s, batch %d" % (patch_size, initial_learning_rate, batch_size))
    network = trainer.TrainerController(patch_size, res_increase, initial_learning_rate, QUICKSAVE, network_name,
                                        low_resblock, hi_resblock)
    network.init_model_dir()
    if restore:
        print("Restoring model %s..." % model_file)
        network.restore_model(model_dir, model_file)
        print("Learning rate", network.optimizer.lr)
    network.train_network(trainset, valset, n_epoch=epochs, testset=testset)
