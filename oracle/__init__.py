"""CPU oracle for the NewtonNet per-edge message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``newtonnet_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``oracle.newtonnet_ref`` is checked (tests/test_oracle.py)
against
  * golden vectors produced by importing the reference itself in the build
    container (tests/golden/gen_golden.py -> tests/golden/*.npz), and
  * the reference's only known-answer data: the 201 frames of
    scripts/md17_md/md.traj (K1) and the `final` row of
    scripts/md17_model/training_1/log.csv (K2)  [SURVEY.md section 4].
"""
