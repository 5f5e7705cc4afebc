"""TEST INFRASTRUCTURE ONLY -- the parity oracle for the UNet_Nested hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline -- never as
the thing shipped.  The product package
(``unet_nested4tiny_objects_keypoints_amd``) never imports this package and
raises if its HIP library is missing.

Pinning: the reference's own tests hold no golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by outputs of the reference
itself, produced in the build container by ``tests/golden/make_golden.py``
(which imports ``/root/reference/models/unet.py`` as-is) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` replays them.
"""
