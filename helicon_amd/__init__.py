"""helicon_amd — MI355X-native denovo3D (twist, rise, Csym) sweep behind jianglab/helicon's
Python signatures.  Hand-written gfx950 kernels in ``csrc/``, reached through the C ABI of
``include/helicon_hip.h``; importing this package does not touch the GPU."""
from .grid import (CandidateGrid, build_grid, layer_line_mask, radial_band_mask, set_to_periodic_range,
                   shard_bounds, sweep_axis)
from .denovo3D import (SweepEngine, SweepResult, apply_helical_symmetry, auto_horizontalize, compute_power_spectra,
                       cosine_similarity, cross_correlation_coefficient, down_scale,
                       estimate_helix_rotation_center_diameter, is_vertical, low_high_pass_filter, process_one_task,
                       rotate_shift_image, simulate_helical_projection, sweep, threshold_data, transform_image,
                       transform_map)
from ._lib import HeliconHipError
from .solver import lsq_reconstruct, lsq_reconstruct_batch

__version__ = "0.1.0"
