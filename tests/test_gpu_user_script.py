"""A whole user script in the reference's style (tools/user_script_smoke.py): default arguments wherever the reference has
them, lasers, the moving window, device-native HDF5 writers and a density diagnostic next to a plain mirror-reading callback, a
restart dump -- 1200 steps of a small laser-target run with the file contents, the heating and the charge bookkeeping checked at
the end.  The reference's integration tests are of this kind (`tests/test_laser_target.py:71-75`: "runs")."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu


def test_reference_style_user_script(capsys):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runpy.run_path(os.path.join(root, "tools", "user_script_smoke.py"), run_name="__main__")
    assert "user script ok" in capsys.readouterr().out
