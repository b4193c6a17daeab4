#!/usr/bin/env python3
"""Generates the small fixtures under tests/golden/ from the data files the reference ships
(tests/golden/data = /root/reference/data, copied verbatim) with the CPU oracle.

The reference's arithmetic (DOLFIN/PETSc) cannot be imported or built here, so these are the
known-answer set-ups of SURVEY.md Appendix C evaluated by the oracle:
  config1_marked_facets.npz  vertex triples of the facets marked 1 / 2 in mesh_facet_region.xml under
                             lexicographic facet numbering (C1) — all on z=0 / z=20
  config1_solution.npy       discrete solution of data/TestHeatTransfer.json by sparse LU, the
                             reference's default solve (C2); equals 350 - 2.5 z to 3e-12
  cube8_cg_history.npy       ||r_k||^2 of Jacobi-PCG (single-reduction recurrence) on the n=8 cube of the
                             config-2 family: an iteration-by-iteration anchor for the HIP solver
Run from the repo root:  python tests/golden/make_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import fem_oracle as fo  # noqa: E402

D = os.path.join(HERE, "data")
co, ce = fo.read_dolfin_xml_mesh(os.path.join(D, "mesh.xml"))
_, fm = fo.read_dolfin_xml_meshfunction(os.path.join(D, "mesh_facet_region.xml"))
facets, _, _ = fo.facet_numbering(ce)
np.savez_compressed(os.path.join(HERE, "config1_marked_facets.npz"), id1=facets[fm == 1], id2=facets[fm == 2])
A = fo.assemble_p1_scalar(co, ce, 20.0)
d1, d2 = fo.dirichlet_dofs_p1(facets, fm, 1), fo.dirichlet_dofs_p1(facets, fm, 2)
Ab, bb = fo.apply_dirichlet(A, np.zeros(len(co)), np.concatenate([d1, d2]),
                            np.concatenate([np.full(len(d1), 350.0), np.full(len(d2), 300.0)]), True)
T = fo.solve_direct(Ab, bb)
assert np.abs(T - (350 - 2.5 * co[:, 2])).max() < 1e-9
np.save(os.path.join(HERE, "config1_solution.npy"), T)
P = fo.heat_box_problem(8)
x, it, hist = fo.pcg_jacobi_single_reduction(P["A"], P["b"], rtol=1e-8)
np.save(os.path.join(HERE, "cube8_cg_history.npy"), np.asarray(hist))
print("goldens written; config1 max|T-(350-2.5z)| = %.2e, cube8 iterations = %d" % (np.abs(T - (350 - 2.5 * co[:, 2])).max(), it))
