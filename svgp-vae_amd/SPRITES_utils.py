"""Host-side SPRITES utilities mirroring the reference's `SPRITES_utils.py` (names and semantics; numpy, no TFRecords):

  sprites_PCA_init(path_train_dict, m, L_action, L_character, seed, N_action)   SPRITES_utils.py:217-279
  aux_data_sprites_utils(batch_size, N, repeats)                                SPRITES_utils.py:317-332 (sprites.py)
  forward_pass_pretraining_repr_NN(frames, labels, repr_NN, classification_layer, test_pipeline)   :335-368 (sprites.py)

The reference reads `sprites_train_dict.p` (frames (N,64,64,3), aux_data (N,2) = [character id, action id]); that file is built
from an external repository and is not obtainable offline, so `path_train_dict` may also be the dict itself (the SPRITES
driver passes its loaded / synthetic training set).
"""
import pickle

import numpy as np

from .sprites import (aux_data_sprites_utils, forward_pass_pretraining_repr_NN,  # noqa: F401  (the reference's names)
                      repr_NN_classification_layer)


def sprites_PCA_init(path_train_dict, m=15, L_action=6, L_character=16, seed=42, N_action=72):
    """SPRITES_utils.py:217-279.  Returns (GPLVM action init (N_action, L_action), inducing-point init (N_action * m,
    L_action + L_character)):
      * GPLVM action vectors = first L_action principal components of the N_action per-action MEAN frames (:240-250);
      * character part of the inducing points = for every action, m draws per PCA axis from a Gaussian KDE of the first
        L_character principal components of ALL train frames (:254-266) -- with the reference's `seed` re-used for every
        action and axis, so the m draws of an axis are identical across actions;
      * rows of action i: [GPLVM_action[i] tiled m times | character vectors] (:268-274).
    PCA = sklearn.decomposition.PCA as in the reference (full SVD with svd_flip sign convention)."""
    import scipy.stats
    from sklearn.decomposition import PCA
    if isinstance(path_train_dict, dict):
        train_dict = path_train_dict
    else:
        with open(path_train_dict, "rb") as f:
            train_dict = pickle.load(f)
    train_frames = np.asarray(train_dict["frames"])
    train_aux_data = np.asarray(train_dict["aux_data"])
    action_ids_train = {i: [] for i in range(N_action)}
    for i in range(len(train_aux_data)):
        action_ids_train[int(train_aux_data[i, 1])].append(i)
    GPLVM_action = np.array([train_frames[ids].mean(axis=0).reshape(-1) for _, ids in action_ids_train.items()])
    GPLVM_action = PCA(n_components=L_action).fit_transform(GPLVM_action)
    train_frames_PCA = PCA(n_components=L_character).fit_transform(train_frames.reshape(len(train_frames), -1))
    inducing_points = []
    for i in range(len(GPLVM_action)):
        char_vectors = np.array([scipy.stats.gaussian_kde(train_frames_PCA[:, ax]).resample(m, seed=seed).reshape(-1)
                                 for ax in range(L_character)]).T
        inducing_points.append(np.hstack((np.tile(GPLVM_action[i, :], (m, 1)), char_vectors)))
    return GPLVM_action, np.concatenate(inducing_points)
