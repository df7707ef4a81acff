"""Pin the ResNet-50 / transform / metrics oracles against reference outputs
(tests/golden/resnet50.npz, metrics_train.npz from make_golden.py)."""
import os

import numpy as np
import torch

from oracle import metrics_oracle as mo
from oracle import resnet_oracle as ro
from sequoia_pub_amd import synth


def test_resnet50_forward_extract_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "resnet50.npz"))
    sd = ro.init_resnet50_state_dict(seed=99, perturb_bn=True)
    s = sum(float(v.double().sum()) for v in sd.values())
    a = sum(float(v.double().abs().sum()) for v in sd.values())
    np.testing.assert_allclose([s, a], z["param_checksum"], rtol=1e-12)
    torch.set_num_threads(8)
    f224 = ro.embed_patches(sd, synth.patches_u8(0, n_patches=2, size=224), batch=1).numpy()
    np.testing.assert_allclose(f224, z["feat224"], rtol=1e-4, atol=1e-4)
    f256 = ro.embed_patches(sd, synth.patches_u8(1, n_patches=1, size=256), batch=1).numpy()
    np.testing.assert_allclose(f256, z["feat256"], rtol=1e-4, atol=1e-4)
    x = ro.transform_patch_u8(synth.patches_u8(0, n_patches=1, size=224))
    with torch.no_grad():
        _, inter = ro.forward_extract(sd, x, return_intermediates=True)
    np.testing.assert_allclose(inter["maxpool"][0, ::8, ::7, ::7].numpy(), z["ref_maxpool_sample"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(inter["layer1"][0, ::16, ::7, ::7].numpy(), z["ref_layer1_sample"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(inter["layer2"][0, ::32, ::4, ::4].numpy(), z["ref_layer2_sample"], rtol=1e-4, atol=1e-4)


def test_metrics_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "metrics_train.npz"))
    labels, preds = z["labels"], z["preds"]
    assert abs(mo.compute_correlations(labels, preds) - float(z["corr"])) < 1e-12
    assert abs(mo.compute_correlations_vectorised(labels, preds) - float(z["corr"])) < 1e-9
    assert abs(mo.mean_absolute_error(labels, preds) - float(z["mae"])) < 1e-6
    assert abs(mo.smape(labels, preds) - float(z["smape"])) < 1e-3
    assert abs(mo.mse(preds, labels) - float(z["mse"])) < 1e-5
