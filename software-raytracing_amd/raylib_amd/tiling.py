"""Image tiling across ranks (one process per GPU).

The reference parallelises over 8x8-pixel WorkCells handed to a thread pool
(reference render/renderer.cc:21-22,305-334).  Across GPUs the same cells are the
unit: rank r of N renders cells r, r+N, r+2N, ... (row-major cell numbering), which
interleaves cheap and expensive regions evenly, writes them back to back into a
device buffer (64 pixels x RGBA per cell), and one gather per frame brings every
rank's buffer to rank 0, which scatters the cells into the row-major image.
No reduction is involved: pixels are disjoint, and because the RNG stream is keyed
by (seed, pixel, sample) the assembled image is bit-identical to a 1-GPU render.
"""
import numpy as np


def num_cells(w, h):
    return ((w + 7) // 8) * ((h + 7) // 8)


def local_cells(w, h, rank, world):
    n = num_cells(w, h)
    return (n - rank + world - 1) // world if rank < n else 0


def padded_cells(w, h, world):
    """Cells per rank after padding to equal size (gather needs equal shapes)."""
    return (num_cells(w, h) + world - 1) // world


def pixel_index_map(w, h, rank, world):
    """For rank's buffer slot k (cell-major, 64 per cell): the row-major pixel index, or -1."""
    cx_n = (w + 7) // 8
    k = np.arange(local_cells(w, h, rank, world) * 64, dtype=np.int64)
    cell = rank + (k // 64) * world
    p = k % 64
    x = (cell % cx_n) * 8 + (p % 8)
    y = (cell // cx_n) * 8 + (p // 8)
    idx = y * w + x
    idx[(x >= w) | (y >= h)] = -1
    return idx


def extract_cells(image, rank, world):
    """Host-side inverse of assemble: the buffer rank would produce for a given (h, w, 4) image."""
    h, w, _ = image.shape
    idx = pixel_index_map(w, h, rank, world)
    flat = image.reshape(-1, 4)
    out = np.zeros((len(idx), 4), np.float32)
    ok = idx >= 0
    out[ok] = flat[idx[ok]]
    return out


def assemble(w, h, world, buffers):
    """buffers[r]: (>= local_cells*64, 4) array of rank r -> (h, w, 4) image."""
    img = np.zeros((h * w, 4), np.float32)
    for r in range(world):
        idx = pixel_index_map(w, h, r, world)
        ok = idx >= 0
        img[idx[ok]] = np.asarray(buffers[r])[: len(idx)][ok]
    return img.reshape(h, w, 4)


def torch_scatter_plan(w, h, world, device):
    """(src_slots, dst_pixels) index tensors for a gathered [world, padded*64, 4] tensor."""
    import torch
    pad = padded_cells(w, h, world) * 64
    src, dst = [], []
    for r in range(world):
        idx = pixel_index_map(w, h, r, world)
        ok = np.nonzero(idx >= 0)[0]
        src.append(r * pad + ok)
        dst.append(idx[ok])
    return (torch.as_tensor(np.concatenate(src), device=device), torch.as_tensor(np.concatenate(dst), device=device))
