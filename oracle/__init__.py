"""CPU oracle for the JoXSZ per-walker log-posterior hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``joxsz_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU
baseline, never as the thing shipped.

Contents
--------
``pyabel_direct``   restatement of ``abel.direct.direct_transform(...,
                    direction='forward', backend='Python')`` (PyAbel; call site
                    ``/root/reference/joxsz_funcs.py:457``).
``mbproj2_parts``   restatement of the mbproj2 pieces the path calls
                    (``Param.prior``, ``ParamGaussian.prior``,
                    ``utils.projectionVolumeMatrix``, ``CountRate.getCountRate``,
                    ``Band.calcProjProfile``, ``utils.cashLogLikelihood``).
``joxsz_oracle``    restatement of ``joxsz_funcs.py:275-301,321-336,375-407,
                    428-437,439-546`` on plain arrays (numpy + scipy).

Parity pinning status
-----------------------------------------------
* ``joxsz_funcs.py:453-493`` (``get_sz_like``) and the profile components are
  PINNED: ``oracle/make_golden.py`` imports the reference module itself in this
  container and commits its outputs under ``tests/golden/``; the oracle is
  checked against those vectors by ``tests/test_oracle_golden.py``.
* PyAbel and mbproj2 are NOT under ``/root/reference`` and are not installed
  (``requirements.txt`` pins no versions): for the Abel step and the X-ray
  projection / priors the oracle is a restatement of their published
  algorithm and is **parity unpinned**; it is cross-checked by analytic
  known answers only (``tests/test_oracle_properties.py``).
"""
