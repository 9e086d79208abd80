"""A dozen trials of tests/fuzz_parity.py (random resolutions, extents, poses) in the GPU suite."""
import pytest

from tests import fuzz_parity

pytestmark = pytest.mark.gpu


def test_random_volumes_and_poses(hip, orc):
    assert fuzz_parity.run(hip, orc, seed=20261003, trials=12) == []


def test_random_depth_images_with_holes(hip, orc):
    from slambench_amd import kfusion
    assert fuzz_parity.run_maps(kfusion, orc, seed=20261003, trials=8) == []


def test_random_poses_through_the_association_kernel(hip, orc):
    assert fuzz_parity.run_track(hip, orc, seed=20261003, trials=4) == []


def test_integrate_at_the_limits_of_its_shortcuts(hip, orc):
    """Arbitrary rotations, cameras inside / outside / on a face of the volume, intrinsics up to 16x (where the rounding bounds
    of the chunk and batch rejects fail and switch them off), planes, noise and mostly-hole depth images: bit-exact."""
    assert fuzz_parity.run_extreme(hip, orc, seed=3, trials=12) == []
