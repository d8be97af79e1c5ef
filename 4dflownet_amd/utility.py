"""Counterpart of src/Network/utility.py (logging helpers used by TrainerController)."""
from time import time


def calculate_time_elapsed(start):
    """-> (hrs, mins, secs) since `start` (utility.py:9-22)."""
    elapsed = time() - start
    hrs = elapsed // 3600
    mins = (elapsed - hrs * 3600) // 60
    secs = int(elapsed - mins * 60 - hrs * 3600)
    return hrs, mins, secs


def log_to_file(filepath, msg):
    """Append `msg` to `filepath` (utility.py:24-26)."""
    with open(filepath, 'a') as f:
        f.write(msg)
