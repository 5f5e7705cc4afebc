"""Checkpoint compatibility: the reference trainer's .pth / .tar formats round-trip between the
reference-shaped oracle model and the HIP model (CPU only: no kernels run)."""
import torch

from oracle.unet_nested_oracle import UNetNestedOracle
from unet_nested4tiny_objects_keypoints_amd import UNet_Nested, checkpoint


def _models():
    torch.manual_seed(0)
    ref = UNetNestedOracle(in_channels=1, n_classes=4, feature_scale=8)
    hip = UNet_Nested(in_channels=1, n_classes=4, feature_scale=8)
    return ref, hip


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), k


def test_reference_pth_loads_into_hip_model_and_back(tmp_path):
    ref, hip = _models()
    path = checkpoint.save_best(ref, str(tmp_path), 3, 0.25, 0.5)  # what trainer.py:246-249 writes
    assert path.endswith("best_epoch_3_heatmaploss_0.25_landmarkloss_0.5.pth")
    assert checkpoint.resume(hip, path) == 0
    _same(ref.state_dict(), hip.state_dict())
    back = tmp_path / "from_hip.pth"
    torch.save(hip.state_dict(), back)
    ref2, _ = _models()
    ref2.load_state_dict(torch.load(back), strict=True)
    _same(hip.state_dict(), ref2.state_dict())


def test_tar_resume_with_optimizer_state(tmp_path):
    ref, hip = _models()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for p in ref.parameters():
        p.grad = torch.ones_like(p)
    opt_ref.step()
    path = str(tmp_path / "ckpt.tar")
    checkpoint.save_checkpoint(ref, opt_ref, 7, path)
    opt = torch.optim.Adam(hip.parameters(), lr=1e-3)
    assert checkpoint.resume(hip, path, optimizer=opt, resume_opt=True) == 8
    _same(ref.state_dict(), hip.state_dict())
    assert opt.state_dict()["state"].keys() == opt_ref.state_dict()["state"].keys()
    assert checkpoint.resume(hip, path) == 0  # without resume_opt the epoch counter restarts (trainer.py:397,411)


def test_dataparallel_prefix_is_stripped(tmp_path):
    ref, hip = _models()
    wrapped = {"module." + k: v for k, v in ref.state_dict().items()}
    path = str(tmp_path / "dp.pth")
    torch.save(wrapped, path)
    checkpoint.resume(hip, path)
    _same(ref.state_dict(), hip.state_dict())

    class Wrapper(torch.nn.Module):  # what nn.DataParallel looks like to the saver
        def __init__(self, m):
            super().__init__()
            self.module = m
    saved = checkpoint.save_best(Wrapper(hip), str(tmp_path), 0, 1.0, 1.0)
    assert all(not k.startswith("module.") for k in torch.load(saved))
