"""Dense density grid of a conditioned field -- the second consumer of the siren sub-API (counterpart of the reference's
extract_shapes.py:15-78; that file itself imports mrcfile / plyfile, which this image lacks, so its numbers are restated
here, not pinned: the grid arithmetic below is written to give the same sample coordinates).

    sigma = sample_generator(generator, z, voxel_resolution=256)        # (N, N, N) numpy array, sigma[i0, i1, i2]

The field kernel takes the whole grid in a few launches (2^22 points per call) instead of the reference's 64^3 slices."""
import numpy as np
import torch


def create_samples(N=256, voxel_origin=(0, 0, 0), cube_length=2.0):
    """(1, N^3, 3) sample coordinates, the corner of the cube and the voxel pitch.

    Point n of the flat grid has column 2 = n mod N and columns 1, 0 = (n / N) mod N, (n / N^2) mod N evaluated in floating
    point WITHOUT flooring -- the reference's convention (extract_shapes.py:24-27), kept so that grids line up with grids
    extracted by it; each column is then scaled by the pitch and shifted by the cube corner (columns 0 / 2 take the corner's
    components 2 / 0: the corner is symmetric in practice)."""
    corner = np.asarray(voxel_origin, dtype=np.float64) - cube_length / 2
    pitch = cube_length / (N - 1)
    n = torch.arange(N ** 3, dtype=torch.int64)
    nf = n.float()
    pts = torch.empty(N ** 3, 3)
    pts[:, 2] = (n % N).float() * pitch + float(corner[0])
    pts[:, 1] = ((nf / N) % N) * pitch + float(corner[1])
    pts[:, 0] = (((nf / N) / N) % N) * pitch + float(corner[2])
    return pts.unsqueeze(0), corner, pitch


def sample_generator(generator, z, voxel_resolution=256, voxel_origin=(0, 0, 0), cube_length=1.2, psi=0.5, max_points=1 << 22):
    """sigma on the N^3 grid as a (N, N, N) numpy array (extract_shapes.py:41-78; `psi` is accepted and unused there too)."""
    N = int(voxel_resolution)
    pts, _, _ = create_samples(N, voxel_origin, cube_length)
    pts = pts.to(generator.device)
    sig = torch.empty((1, N ** 3), dtype=torch.float32, device=generator.device)
    with torch.no_grad():
        for s0 in range(0, N ** 3, max_points):
            chunk = pts[:, s0:s0 + max_points].contiguous()
            sig[:, s0:s0 + chunk.shape[1]] = generator.siren(chunk, z, N, 1)[..., 3]
    return sig.reshape(N, N, N).cpu().numpy()
