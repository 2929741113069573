"""CPU restatement of the feed/consume steps around the forward.  TEST INFRASTRUCTURE.

``feed`` follows ``TestingDataset.__getitem__`` (``pointstowood/src/predicter.py:78-94``) and the PyG
collation rules used at :177; ``consume`` follows the loop body :199-215 (``nan_to_num`` -> sigmoid ->
``>= is_wood`` -> per-voxel un-shift with ``local_shift[3b:3b+3]``).  Pure tensor code - no third-party
arithmetic is involved, so these are pinned by construction (statement-by-statement restatement).
"""
import numpy as np
import torch


def feed(point_cloud: torch.Tensor, reflectance_index: int = 3):
    pos = torch.as_tensor(point_cloud[:, :3], dtype=torch.float)
    reflectance = torch.as_tensor(point_cloud[:, reflectance_index], dtype=torch.float)
    local_shift = torch.mean(pos[:, :3], axis=0)
    pos = pos - local_shift
    scaling_factor = torch.sqrt((pos ** 2).sum(dim=1)).max()
    nan_mask = torch.isnan(pos).any(dim=1) | torch.isnan(reflectance)
    return {"pos": pos[~nan_mask], "reflectance": reflectance[~nan_mask], "local_shift": local_shift, "sf": scaling_factor}


def consume(logits, pos, batch, local_shift, is_wood):
    outputs = torch.nan_to_num(logits)
    probs = torch.sigmoid(outputs)
    preds = np.expand_dims((probs >= is_wood).type(torch.int64).numpy(), axis=1)
    output = np.concatenate((pos.numpy(), preds, np.expand_dims(probs.numpy(), axis=1)), axis=1)
    out = []
    for b in np.unique(batch.numpy()):
        ob = np.asarray(output[batch.numpy() == b])
        ob[:, :3] = ob[:, :3] + np.asarray(local_shift)[3 * b: 3 + 3 * b]
        out.append(ob)
    return np.vstack(out)
