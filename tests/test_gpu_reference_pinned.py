"""The checks that are pinned to the reference itself or that cover the multi-process path need no GPU and run in the
CPU suite -- which the round-end driver does not run on the GPU box.  This module puts the same checks under the `gpu`
marker as well (it calls the very same test functions), so that the driver's `-m gpu` record lists them:

  * the C++ and the Python outlier injectors against files written by the reference's own scripts/generateDataset.py
    (the one piece of the reference that runs in the authoring container);
  * the row-sharded matrix over a world-size 2 / 3 gloo process group;
  * the oracle against its known-answer tests and the committed expectations.
"""
import pytest

import test_dist_gloo
import test_host_cpp
import test_host_logic
import test_oracle_kat

pytestmark = pytest.mark.gpu

built = test_host_cpp.built                     # the module-scoped build fixture of test_host_cpp


@pytest.mark.parametrize("clean,spoiled,flags", test_host_cpp.GEN_CASES)
def test_cxx_injector_reproduces_the_reference_script_byte_for_byte(built, tmp_path, clean, spoiled, flags):
    test_host_cpp.test_generate_dataset_reproduces_the_reference_script_byte_for_byte(built, tmp_path, clean, spoiled, flags)


@pytest.mark.parametrize("name,clean,spoiled,kw", test_host_logic.CASES)
def test_python_injector_reproduces_the_reference_script(name, clean, spoiled, kw):
    test_host_logic.test_injector_reproduces_reference_script(name, clean, spoiled, kw)


def test_3d_injector_keeps_the_wxyz_quirk():
    test_host_logic.test_3d_injector_keeps_the_wxyz_quirk()


@pytest.mark.parametrize("name,clean,spoiled,kw", test_host_logic.CASES)
def test_oracle_matches_committed_expected(oracle, name, clean, spoiled, kw):
    test_host_logic.test_oracle_matches_committed_expected(oracle, name, clean, spoiled, kw)


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_matrix_gloo(world):
    test_dist_gloo.test_row_sharded_matrix_gloo(world)


@pytest.mark.parametrize("dim", [2, 3])
def test_oracle_closed_form_single_loop_chi2(oracle, dim):
    test_oracle_kat.test_closed_form_single_loop_chi2(oracle, dim)


@pytest.mark.parametrize("dim", [2, 3])
def test_oracle_jacobians_vs_finite_differences(oracle, dim):
    test_oracle_kat.test_jacobians_vs_finite_differences(oracle, dim)
