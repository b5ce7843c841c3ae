"""CPU: programmatic state-dict schemas == schemas captured from the imported reference modules."""
import pytest
import se_amd
from se_amd import schemas
from conftest import load_schema


@pytest.mark.parametrize('name', ['lstm', 'crn', 'dpcrn', 'dccrn', 'fullsubnet', 'fullsubnet_gru', 'gcrn', 'cts_step1', 'cts_step2', 'taylorsenet', 'g2net', 'uformer',
                                  'cts_step1_new', 'cts_step2_new', 'taylorsenet_new', 'g2net_new'])
def test_schema_matches_reference(name):
    ref = load_schema(name)
    mine = schemas.SCHEMAS[name]()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k][0]) == tuple(ref[k][0]) and mine[k][1] == ref[k][1], k
