"""Oracle (TEST INFRASTRUCTURE ONLY): the fixed-instance-count rule of the reference's datasets.

Restates data/dataset.py:151-175 (IvYGAP_Dataset.read_img; the TCGA copy at :383-407 is identical): a bag of num_patches
rows is brought to max_num = args.fixdim rows - shorter bags are repeated floor(max_num / num_patches) times and topped up
with their first max_num % num_patches rows, longer bags keep row int(np.around(i * (num_patches / max_num))).
The reference embeds the rule in its image-reading loop (it needs the patch files to run as it stands).  PINNED: tests/golden/
make_golden.py (case_fixdim) executes the statements of read_img themselves - source taken from the imported class, file access
(np.load / io.imread / os.listdir) replaced by stand-ins that return row numbers - and stores the resulting index vectors for 33
(bag length, fixdim) pairs in tests/golden/fixdim_indices.npz; tests/test_bag_store.py checks this restatement, the package's
vectorised form and the device kernel against them, bit for bit."""
from __future__ import annotations

import numpy as np


def fixdim_indices(num_patches: int, max_num: int) -> np.ndarray:
    """Source row of every one of the max_num output rows."""
    use = num_patches if num_patches <= max_num else max_num
    if num_patches <= max_num:
        times = int(np.floor(max_num / num_patches))            # :155
        remaining = max_num % num_patches                       # :156
        ori = list(range(use))
        out = list(ori)
        if times > 1:                                           # :164-167
            for _ in range(times - 1):
                out = out + ori
        if not remaining == 0:                                  # :168-169
            out = out + ori[0:remaining]
    else:
        out = [int(np.around(i * (num_patches / max_num))) for i in range(use)]   # :172-173
    return np.asarray(out, dtype=np.int64)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) as uint16; NaN stays NaN."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        r = np.where(nan, ((u >> 16) | 0x0040).astype(np.uint16), r)
    return r
