"""The train.py counterpart end to end on the GPU: a few iterations, validation with argmax masks,
checkpoint in the reference's dictionary format, and resume (train.py:525-609, 972-1016)."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_validate_checkpoint_resume(tmp_path, capsys):
    if not torch.cuda.is_available():
        pytest.fail("needs a GPU")
    from iswm_amd import train
    ck = str(tmp_path / "ck")
    base = ["--model", "deeplabv3plus_resnet50", "--crop_size", "65", "--batch_size", "4", "--synthetic_len", "16",
            "--optimizer", "sgd", "--loss_type", "IWce_loss", "--print_interval", "2", "--val_interval", "2",
            "--val_batch_size", "4", "--checkpoints_dir", ck, "--num_workers", "0"]
    train.main(base + ["--total_itrs", "4"])
    out = capsys.readouterr().out
    assert "Itrs 4/4" in out and "Validation @2" in out
    files = glob.glob(os.path.join(ck, "best_*.pth"))
    assert len(files) == 1
    ckpt = torch.load(files[0], map_location="cpu", weights_only=True)
    for key in ("model_state", "optimizer_state", "scheduler_state", "cur_itrs", "best_score", "model_config"):
        assert key in ckpt
    assert len(ckpt["model_state"]) == 374
    assert ckpt["model_config"]["model_name"] == "deeplabv3plus_resnet50"
    train.main(base + ["--total_itrs", "6", "--ckpt", files[0], "--continue_training"])
    out = capsys.readouterr().out
    assert "Model restored" in out and "Itrs 6/6" in out


def test_train_with_device_augmentation(tmp_path, capsys):
    """--device_augment: uint8 tiles -> one augmentation kernel per batch -> the same training loop; validation
    goes through the fused argmax + confusion-matrix kernel"""
    from iswm_amd import train
    train.main(["--model", "deeplabv3plus_resnet50", "--crop_size", "65", "--batch_size", "4", "--synthetic_len", "16",
                "--optimizer", "sgd", "--loss_type", "IWce_loss", "--print_interval", "2", "--val_interval", "3",
                "--val_batch_size", "4", "--checkpoints_dir", str(tmp_path / "ck"), "--total_itrs", "3",
                "--device_augment", "--num_workers", "2"])
    out = capsys.readouterr().out
    assert "Itrs 2/3" in out and "Validation @3" in out and "MIoU" in out


@pytest.mark.parametrize("optimizer,loss_type", [("adam", "ce_loss"), ("adamw", "IWce_loss")])
def test_train_other_optimizers_and_losses_resume(tmp_path, capsys, optimizer, loss_type):
    """setup_optimizer / setup_criterion branches of train.py:421-459 (Adam, AdamW with weight decay; unweighted CE):
    train, checkpoint, resume with the optimizer state restored into the flat arena"""
    from iswm_amd import train
    ck = str(tmp_path / "ck")
    base = ["--model", "deeplabv3plus_resnet50", "--crop_size", "65", "--batch_size", "4", "--synthetic_len", "16",
            "--optimizer", optimizer, "--loss_type", loss_type, "--print_interval", "2", "--val_interval", "2",
            "--val_batch_size", "4", "--checkpoints_dir", ck, "--num_workers", "0"]
    train.main(base + ["--total_itrs", "2"])
    out = capsys.readouterr().out
    assert "Itrs 2/2" in out and "Validation @2" in out
    files = glob.glob(os.path.join(ck, "best_*.pth"))
    assert len(files) == 1
    ckpt = torch.load(files[0], map_location="cpu", weights_only=True)
    st = ckpt["optimizer_state"]["state"]
    assert len(st) > 0 and all("exp_avg" in v and "exp_avg_sq" in v for v in st.values())
    train.main(base + ["--total_itrs", "4", "--ckpt", files[0], "--continue_training"])
    out = capsys.readouterr().out
    assert "Model restored" in out and "Itrs 4/4" in out
