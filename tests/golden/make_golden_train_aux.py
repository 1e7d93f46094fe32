"""Fixture for the inducing-point initialiser: what is needed to rebuild the 4 050 train aux rows of the reference's default
dataset without the (missing) train_data3.p -- the train-id mask and the test angle from the reference's own data files.
Run in the build container (needs /root/reference): python tests/golden/make_golden_train_aux.py"""
import os
import pickle

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/MNIST data/"
mask = np.asarray(pickle.load(open(REF_DATA + "train_ids_mask3.p", "rb")), dtype=bool)
test_angle = float(pickle.load(open(REF_DATA + "test_data3.p", "rb"))["aux_data"][0, 1])
assert mask.shape == (5400,) and int(mask.sum()) == 4050
np.savez_compressed(os.path.join(HERE, "mnist_train_ids_mask.npz"), train_ids_mask=mask, test_angle=np.float64(test_angle))
print("wrote mnist_train_ids_mask.npz", mask.sum(), test_angle)
