"""Launch knobs (ultra_set_tuning / ultra_get_tuning) from Python: scoped changes restore what was there.  No GPU needed."""
from ultra_amd import rspmm


def test_tuning_scope_changes_only_what_it_names_and_restores():
    rspmm.set_tuning()
    base = rspmm.get_tuning()
    assert base["grid"] == 0 and base["update_form"] == 0
    rspmm.set_tuning(update_form=1, unroll=4)
    try:
        with rspmm.tuning_scope(grid=192):
            inner = rspmm.get_tuning()
            assert inner["grid"] == 192 and inner["update_form"] == 1 and inner["unroll"] == 4
            with rspmm.tuning_scope(update_form=3, grid=0):
                assert rspmm.get_tuning()["update_form"] == 3 and rspmm.get_tuning()["grid"] == 0
            assert rspmm.get_tuning() == inner
        after = rspmm.get_tuning()
        assert after["grid"] == 0 and after["update_form"] == 1 and after["unroll"] == 4
        try:
            with rspmm.tuning_scope(grid=64):
                raise ValueError("leave through an exception")
        except ValueError:
            pass
        assert rspmm.get_tuning() == after
    finally:
        rspmm.set_tuning()
    assert rspmm.get_tuning() == base


def test_empty_scope_is_a_no_op():
    rspmm.set_tuning()
    with rspmm.tuning_scope():
        assert rspmm.get_tuning()["grid"] == 0
    assert rspmm.get_tuning()["grid"] == 0
