"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the bundle-adjustment hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / CPU baseline -- never as the thing that
is measured or shipped.  The product (``structure_from_motion_amd``) never
imports this package and fails loudly when its HIP library is missing.

Parity status
-------------
* Solver / objective (scipy ``least_squares`` configuration of
  ``/root/reference/src/BundleAdjustment.py:210-220``): pinned by golden
  vectors generated from the reference's own ``global_bundle_adjustment`` in
  the build container (``tests/golden/make_golden.py``).
* OpenCV arithmetic (``cv2.Rodrigues``, ``cv2.projectPoints``; reference call
  sites ``BundleAdjustment.py:57,82,113,142``): **parity unpinned**.  OpenCV is
  a third-party dependency of the reference (``opencv-python``, version not
  pinned upstream, ``README.md:60``), absent from this image; the reference
  holds no test or fixture for it.  ``oracle/cv2_standin.py`` restates
  OpenCV's published formulas.
"""
