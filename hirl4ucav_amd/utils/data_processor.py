"""Expert-data CSV I/O in the reference's on-disk format (hirl/utils/data_processor.py:5-21): two rows, row 0 the
stringified state arrays, row 1 the stringified action arrays, parsed with np.fromstring(item[1:-1], sep=' ')."""
import csv

import numpy as np


def _parse(cell):
    return np.array(cell.strip()[1:-1].replace("\n", " ").split(), dtype=np.float64)


def read_data(data_dir):
    with open(data_dir, newline="") as f:
        rows = list(csv.reader(f))
    npstate = np.array([_parse(c) for c in rows[0]])
    npaction = np.array([_parse(c) for c in rows[1]])
    return npstate, npaction


def write_data(npstate, npaction, data_dir):
    with open(data_dir, "w", newline="") as f:
        csv.writer(f).writerows([npstate, npaction])


def up_sample(BCActions):
    return np.where(BCActions[:, 3] == 1)[0]
