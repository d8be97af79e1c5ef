"""Counterparts of src/Network/h5util.py and src/utils/prediction_utils.py: append-along-axis-0 HDF5 writers
(float64 is stored as float32, optional gzip), on the built-in HDF5 writer of h5io.py when h5py is absent."""
import os

from . import h5io


def save_to_h5(output_filepath, col_name, dataset, compression=None):
    """prediction_utils.save_to_h5 (prediction_utils.py:15-28)."""
    h5io.append_dataset(output_filepath, col_name, dataset, compression=compression)


def save_predictions(output_path, output_filename, col_name, dataset, compression=None):
    """h5util.save_predictions (h5util.py:5-23): one dataset, appended along axis 0."""
    if not os.path.isdir(output_path):
        os.makedirs(output_path)
    h5io.append_dataset(os.path.join(output_path, output_filename), col_name, dataset, compression=compression)


def save_prediction_columns(output_dir, output_filename, colnames, predictions, compression=None):
    """prediction_utils.save_predictions (prediction_utils.py:5-12): predictions (N,X,Y,Z,len(colnames)), one dataset per column."""
    if not os.path.isdir(output_dir):
        os.makedirs(output_dir)
    output_filepath = os.path.join(output_dir, output_filename)
    for i, col in enumerate(colnames):
        save_to_h5(output_filepath, col, predictions[:, :, :, :, i], compression=compression)
    print("Prediction saved to %s" % output_filepath)
