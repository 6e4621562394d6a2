#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

Imports ``/root/reference/simulators`` by path and pushes seeded inputs through its own ``Gate.apply`` /
``Simulator.run`` / ``cv_simulator`` helpers; what is written are inputs and the reference's outputs -- data,
no reference code.  The reference does not exist on the GPU box: tests only read the committed ``.npz`` files.

    python tests/golden/generate_golden.py          # rewrites tests/golden/*.npz

Covers SURVEY.md section 8c items (1)-(7).
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REFERENCE = Path("/root/reference")
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(REFERENCE))

from fixture_io import cv_mps_program, gkp_programs, pack_ops  # noqa: E402
from quantum_computations_amd import workloads as W  # noqa: E402

# --- the reference (untrusted public code: imported to be RUN, nothing is copied) ---------------------------
from simulators.dv_simulator import gates as ref_gates  # noqa: E402
from simulators.dv_simulator import numpy_quantum as ref_npq  # noqa: E402
from simulators.dv_simulator.simulator import ClassicalControl as RefControl  # noqa: E402
from simulators.dv_simulator.simulator import Simulator as RefSimulator  # noqa: E402
from simulators.dv_simulator.states import State as RefState  # noqa: E402
from simulators.cv_simulator import gates as ref_cv  # noqa: E402
from simulators.cv_simulator import utils as ref_cvu  # noqa: E402
from simulators.cv_simulator.mps import MPS as RefMPS  # noqa: E402

NAMED = ("I", "X", "Y", "Z", "H", "P", "Pdg", "T", "Tdg", "CX", "CZ", "SWAP")


def ref_gate(o: dict):
    name, idx = o["name"], o["indices"]
    if name in NAMED:
        g = getattr(ref_gates, name)(*idx)
    elif name == "RZ":
        g = ref_gates.RZ(idx[0], o["angle"])
    elif name == "M":
        g = ref_gates.M(idx[0], o["theta"], o["phi"], result=o.get("result"))
    elif name == "Insert":
        g = ref_gates.Insert(idx[0], RefState[o["state"]])
    else:
        g = ref_gates.Gate(list(idx), np.asarray(o["matrix"]))
    ctl = o.get("control")
    if ctl is not None:
        g = RefControl(g, list(ctl.get("pos", [])), list(ctl.get("neg", [])))
    return g


def ref_run(ops, state):
    sim = RefSimulator([ref_gate(o) for o in ops])
    out = sim.run(state)
    return out, sim.results


def complex_ket(n, rng):
    v = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    return v / np.linalg.norm(v)


def save(name, **arrays):
    path = HERE / name
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {path.stat().st_size / 1024:.1f} KiB")


# (1) every gate class on every position, n in {1, 2, 3, 5} --------------------------------------------------
def gen_single_gates():
    rng = np.random.default_rng(1)
    out = {}
    cases = []
    for n in (1, 2, 3, 5):
        ket = complex_ket(n, rng)
        out[f"in_n{n}"] = ket
        results = []
        for name in ("I", "X", "Y", "Z", "H", "P", "Pdg", "T", "Tdg"):
            for q in range(n):
                o = W.op(name, q)
                results.append(ref_gate(o).apply(ket))
                cases.append({"n": n, "name": name, "indices": [q], "row": len(results) - 1})
        for q in range(n):
            angle = float(rng.uniform(-np.pi, np.pi))
            results.append(ref_gates.RZ(q, angle).apply(ket))
            cases.append({"n": n, "name": "RZ", "indices": [q], "angle": angle, "row": len(results) - 1})
        for name in ("CX", "CZ", "SWAP"):
            for q0 in range(n):
                for q1 in range(n):
                    if q0 != q1:
                        results.append(getattr(ref_gates, name)(q0, q1).apply(ket))
                        cases.append({"n": n, "name": name, "indices": [q0, q1], "row": len(results) - 1})
        out[f"out_n{n}"] = np.stack([np.asarray(r, dtype=np.complex128) for r in results])
    # real-ket dtype behaviour (H on a real ket stays float64, T makes it complex128)
    real_ket = np.array([0.6, 0.0, 0.8, 0.0])
    out["real_in"] = real_ket
    out["real_H1"] = ref_gates.H(1).apply(real_ket)
    out["real_T0"] = ref_gates.T(0).apply(real_ket)
    out["int_X0"] = ref_gates.X(0).apply(np.array([1, 0, 0, 0]))
    save("dv_single_gates.npz", cases=json.dumps(cases), **out)


# (2) expand_gate / permute_tensor_product tables -------------------------------------------------------------
def gen_expand_gate():
    rng = np.random.default_rng(2)
    u2, u4 = W.haar_unitary(2, rng), W.haar_unitary(4, rng)
    arrays, cases = {"u2": u2, "u4": u4}, []
    for label, m, N, targets in [("cx_n3_t20", ref_npq.CX, 3, [2, 0]), ("cx_n3_t01", ref_npq.CX, 3, [0, 1]),
                                 ("u2_n3_t1", u2, 3, [1]), ("u4_n4_t31", u4, 4, [3, 1]),
                                 ("u4_n4_t02", u4, 4, [0, 2]), ("swap_n4_t03", ref_npq.SWAP, 4, [0, 3])]:
        arrays[label] = ref_npq.expand_gate(np.asarray(m), N, list(targets))
        cases.append({"label": label, "N": N, "targets": targets,
                      "matrix": "u2" if m is u2 else "u4" if m is u4 else label.split("_")[0].upper()})
    ket = complex_ket(4, rng)
    arrays["perm_in"] = ket
    for label, order in [("perm_1302", [1, 3, 0, 2]), ("perm_3210", [3, 2, 1, 0]), ("perm_0123", [0, 1, 2, 3])]:
        arrays[label] = ref_npq.permute_tensor_product(ket, order)
        cases.append({"label": label, "order": order})
    op = ref_npq.expand_gate(u4, 3, [0, 2])
    arrays["perm_op_in"] = op
    arrays["perm_op_201"] = ref_npq.permute_tensor_product(op, [2, 0, 1])
    save("dv_expand_gate.npz", cases=json.dumps(cases), **arrays)


# (3) cfg1: 4-qubit random Clifford circuits -------------------------------------------------------------------
def gen_clifford():
    arrays, seeds = {}, list(range(6))
    for seed in seeds:
        ops = W.random_clifford_circuit(4, 20, seed)
        final, _ = ref_run(ops, [RefState.ZERO] * 4)
        meta, mats = pack_ops(ops)
        arrays[f"meta_{seed}"], arrays[f"mats_{seed}"] = meta, mats
        arrays[f"final_{seed}"] = np.asarray(final)
    save("dv_clifford_n4.npz", seeds=np.array(seeds), **arrays)


# (4) depth-100 random circuits at n in {6, 8, 10} (the cfg2 generator at sizes the reference can run) ---------
def gen_random_circuits():
    arrays, cases = {}, []
    for n, depth, seed in [(6, 100, 106), (8, 100, 108), (10, 100, 110), (7, 60, 7), (9, 40, 9)]:
        ops = W.random_circuit(n, depth, seed)
        init = W.random_ket(n, seed)
        final, _ = ref_run(ops, init)
        tag = f"n{n}_s{seed}"
        arrays[f"meta_{tag}"], arrays[f"mats_{tag}"] = pack_ops(ops)
        arrays[f"init_{tag}"], arrays[f"final_{tag}"] = init, np.asarray(final, dtype=np.complex128)
        cases.append({"tag": tag, "n": n, "depth": depth, "seed": seed})
        print(f"  random circuit n={n} depth={depth} done")
    save("dv_random_circuits.npz", cases=json.dumps(cases), **arrays)


# (5) measurement, insertion, classical control, density matrices -----------------------------------------------
def gen_measure_insert():
    rng = np.random.default_rng(5)
    arrays, cases = {}, []
    for n in (1, 3, 4):
        ket = complex_ket(n, rng)
        arrays[f"ket_n{n}"] = ket
        for label, theta, phi in [("MZ", 0.0, 0.0), ("MX", np.pi / 2, 0.0), ("Mgen", 0.7, 1.3), ("Mgen2", 2.1, -0.4)]:
            for q in range(n):
                for result in (0, 1):
                    gate = ref_gates.M(q, theta, phi, result=result)
                    out, s = gate.apply(ket)
                    # branch norms exactly as gates.py:173-181 computes them, with the reference's primitives
                    rot = ref_npq.axis_rotation(phi, [0, 0, 1]) @ ref_npq.axis_rotation(theta, [0, 1, 0])
                    ops = [ref_npq.IDTY] * n
                    ops[q] = (rot @ ref_npq.ZERO, rot @ ref_npq.ONE)[result]
                    norm = ref_npq.norm(ref_npq.tensor(*ops) @ ket)
                    key = f"{label}_n{n}_q{q}_r{result}"
                    arrays[key] = np.asarray(out, dtype=np.complex128)
                    cases.append({"kind": "measure", "key": key, "n": n, "q": q, "theta": theta, "phi": phi,
                                  "result": result, "norm": float(norm), "s": int(s)})
    # insert chains from the empty register
    chains = {
        "ins_a": [(0, "ZERO"), (1, "PLUS"), (1, "ONE")],
        "ins_b": [(0, "T"), (0, "H"), (1, "MINUS"), (3, "TDG"), (2, "ONE")],
        "ins_c": [(0, "PLUS"), (0, "PLUS"), (2, "T")],
    }
    for key, chain in chains.items():
        ops = [{"name": "Insert", "indices": [q], "matrix": None, "state": s} for q, s in chain]
        out, _ = ref_run(ops, None)
        arrays[key] = np.asarray(out, dtype=np.complex128)
        cases.append({"kind": "insert_chain", "key": key, "chain": chain})
    # insertion into a random register at every position
    ket3 = arrays["ket_n3"]
    for q in range(4):
        key = f"ins_n3_q{q}"
        arrays[key] = np.asarray(ref_gates.Insert(q, RefState.T).apply(ket3), dtype=np.complex128)
        cases.append({"kind": "insert", "key": key, "n": 3, "q": q, "state": "T"})
    # classical control: teleportation-style circuit with forced outcomes
    for r0 in (0, 1):
        for r1 in (0, 1):
            ops = [
                {"name": "Insert", "indices": [0], "matrix": None, "state": "T"},
                {"name": "Insert", "indices": [1], "matrix": None, "state": "ZERO"},
                {"name": "Insert", "indices": [2], "matrix": None, "state": "ZERO"},
                W.op("H", 1), W.op("CX", 1, 2), W.op("CX", 0, 1), W.op("H", 0),
                {"name": "M", "indices": [0], "matrix": None, "theta": 0.0, "phi": 0.0, "result": r0},
                {"name": "M", "indices": [0], "matrix": None, "theta": 0.0, "phi": 0.0, "result": r1},
                {**W.op("X", 0), "control": {"pos": [1], "neg": []}},
                {**W.op("Z", 0), "control": {"pos": [0], "neg": []}},
                {**W.op("H", 0), "control": {"pos": [], "neg": [0, 1]}},
            ]
            out, results = ref_run(ops, None)
            key = f"ctl_{r0}{r1}"
            arrays[key] = np.asarray(out, dtype=np.complex128)
            arrays[f"{key}_meta"], arrays[f"{key}_mats"] = pack_ops(ops)
            cases.append({"kind": "control", "key": key, "results": [int(r) for r in results]})
    # density matrices, n <= 3
    for n in (1, 2, 3):
        ket = complex_ket(n, rng)
        rho = np.outer(ket, ket.conj())
        mix = 0.7 * rho + 0.3 * np.identity(1 << n) / (1 << n)
        arrays[f"rho_n{n}"] = mix
        gates = [W.op("H", 0), W.op("T", n - 1), W.op("U", 0, matrix=W.haar_unitary(2, rng))]
        if n >= 2:
            gates += [W.op("CX", n - 1, 0), W.op("U", 0, n - 1, matrix=W.haar_unitary(4, rng))]
        for j, o in enumerate(gates):
            key = f"rho_n{n}_g{j}"
            arrays[key] = np.asarray(ref_gate(o).apply(mix), dtype=np.complex128)
            arrays[f"{key}_meta"], arrays[f"{key}_mats"] = pack_ops([o])
            cases.append({"kind": "density", "key": key, "n": n})
    save("dv_measure_insert.npz", cases=json.dumps(cases), **arrays)


# (6) the 3-qubit Grover anchor and the hand-decomposed CCZ ---------------------------------------------------
def gen_readout():
    """numpy_quantum.py:110-166: ket2dm, fidelity in its four branches, purity -- on seeded kets and mixed states; plus
    reduced density matrices obtained from the reference's own ket2dm by summing out the other qubits."""
    rng = np.random.default_rng(148)
    cases, arrays = [], {}
    for n in (2, 3, 5):
        a, b = complex_ket(n, rng), complex_ket(n, rng)
        mix = rng.random(3)
        mix /= mix.sum()
        kets = [complex_ket(n, rng) for _ in range(3)]
        rho = sum(w * ref_npq.ket2dm(k) for w, k in zip(mix, kets))
        sigma = 0.7 * ref_npq.ket2dm(a) + 0.3 * np.identity(1 << n) / (1 << n)
        tag = f"n{n}"
        arrays.update({f"{tag}_a": a, f"{tag}_b": b, f"{tag}_rho": rho, f"{tag}_sigma": sigma})
        values = {"ket_ket": float(ref_npq.fidelity(a, b)), "ket_dm": float(ref_npq.fidelity(a, rho)),
                  "dm_ket": float(ref_npq.fidelity(sigma, b)), "dm_dm": float(ref_npq.fidelity(rho, sigma)),
                  "purity_rho": float(ref_npq.purity(rho)), "purity_sigma": float(ref_npq.purity(sigma)),
                  "purity_pure": float(ref_npq.purity(ref_npq.ket2dm(a)))}
        kept_sets = [[0], [n - 1], [1, 0]] + ([[2, 0, 4], [4, 3, 2, 1]] if n == 5 else [])
        for kept in kept_sets:
            full = ref_npq.ket2dm(a).reshape((2,) * (2 * n))
            rest = [q for q in range(n) if q not in kept]
            # rho_kept[i, j] = sum_rest full[i, rest, j, rest], kept[0] the most significant bit of i and j
            sub = "".join(chr(97 + q) for q in range(n)) + "".join(chr(97 + q) if q in rest else chr(65 + q) for q in range(n))
            out = "".join(chr(97 + q) for q in kept) + "".join(chr(65 + q) for q in kept)
            arrays[f"{tag}_rdm_{'_'.join(map(str, kept))}"] = np.einsum(f"{sub}->{out}", full).reshape(1 << len(kept), -1)
        cases.append({"n": n, "tag": tag, "values": values, "kept": kept_sets})
    save("dv_readout.npz", cases=json.dumps(cases), **arrays)


def gen_grover():
    arrays, cases = {}, []
    for tagged in ([3, 6], [0, 4], [2, 7]):
        ops = W.grover3_ops(tagged)
        out, _ = ref_run(ops, None)
        key = "grover_" + "".join(map(str, tagged))
        arrays[key] = np.asarray(out, dtype=np.complex128)
        cases.append({"key": key, "tagged": tagged})
    ccz = np.zeros((8, 8), dtype=np.complex128)
    for col in range(8):
        out, _ = ref_run(W.ccz_ops(), ref_npq.basis_state(col, 3))
        ccz[:, col] = out
    arrays["ccz_operator"] = ccz
    save("dv_grover3.npz", cases=json.dumps(cases), **arrays)


# (7) cv_simulator: the d x d / (d, d)-plane linear maps behind the gates, and short dense sequences -----------
def gen_cv():
    arrays, cases = {}, []
    d = 32
    qs = np.linspace(-8.0, 8.0, d)
    arrays["qs32"] = qs
    eye = np.identity(d, dtype=np.complex128)

    def single_mode_matrix(gate):
        # column j of the operator = the gate applied to the j-th grid basis vector on a 1-mode MPS
        cols = []
        for j in range(d):
            mps = RefMPS(qs, [eye[:, j].copy()])
            gate.apply(mps, rng=None)
            cols.append(mps[0].reshape(-1))
        return np.stack(cols, axis=1)

    for label, gate in [("X_0.7", ref_cv.X(0, 0.7)), ("X_1.3_dag", ref_cv.X(0, 1.3, dagger=True)),
                        ("F", ref_cv.F(0)), ("F_dag", ref_cv.F(0, dagger=True)),
                        ("Z_0.9", ref_cv.Z(0, 0.9)), ("P_0.5", ref_cv.P(0, 0.5)),
                        ("P_0.5_dag", ref_cv.P(0, 0.5, dagger=True))]:
        arrays["cv1_" + label] = single_mode_matrix(gate)
        cases.append({"kind": "single", "key": "cv1_" + label})
    for theta in (0.4, -1.1, np.pi / 2):
        key = f"cv1_rotation_{theta:.3f}"
        arrays[key] = ref_cvu.rotation(qs, eye, theta, axis=0)
        cases.append({"kind": "rotation", "key": key, "theta": float(theta)})

    # two-mode gates on small grids: the full d^2 x d^2 operator, truncation disabled
    d2 = 8
    qs2 = np.linspace(-4.0, 4.0, d2)
    arrays["qs8"] = qs2
    eye2 = np.identity(d2, dtype=np.complex128)
    exact = {"rel_err": 0.0, "abs_err": 0.0}

    def two_mode_operator(make_gate):
        op = np.zeros((d2 * d2, d2 * d2), dtype=np.complex128)
        for j0 in range(d2):
            for j1 in range(d2):
                mps = RefMPS(qs2, [eye2[:, j0].copy(), eye2[:, j1].copy()])
                make_gate().apply(mps, rng=None)
                op[:, j0 * d2 + j1] = mps.contract().reshape(-1)
        return op

    two_mode = {
        "cv2_CZ_0.8": lambda: ref_cv.CZ(0, 1, 0.8, **exact),
        "cv2_CZ_0.8_dag": lambda: ref_cv.CZ(0, 1, 0.8, dagger=True, **exact),
        "cv2_BS_pi4": lambda: ref_cv.BS(0, 1, np.pi / 4, **exact),
        "cv2_BS_0.3_rev": lambda: ref_cv.BS(1, 0, 0.3, **exact),
        "cv2_BS_0.3_dag": lambda: ref_cv.BS(0, 1, 0.3, dagger=True, **exact),
        "cv2_CX_1.0": lambda: ref_cv.CX(0, 1, 1.0, **exact),
        "cv2_CX_1.0_rev": lambda: ref_cv.CX(1, 0, 1.0, **exact),
        "cv2_SWAP": lambda: ref_cv.SWAP(0, 1, **exact),
    }
    for key, make in two_mode.items():
        arrays[key] = two_mode_operator(make)
        cases.append({"kind": "two_mode", "key": key})

    # a short 3-mode sequence on random product-free input, dense result via MPS.contract()
    rng = np.random.default_rng(7)
    d3 = 12
    qs3 = np.linspace(-5.0, 5.0, d3)
    arrays["qs12"] = qs3
    psi = rng.standard_normal((d3, d3, d3)) + 1j * rng.standard_normal((d3, d3, d3))
    psi /= np.linalg.norm(psi)
    arrays["cv_seq_in"] = psi
    # exact MPS of psi by two SVD splits (rank is at most d3 and d3, nothing truncated)
    u, s, vh = np.linalg.svd(psi.reshape(d3, d3 * d3), full_matrices=False)
    a0 = (u * s).reshape(1, d3, -1)
    rest = vh.reshape(-1, d3, d3)
    u2, s2, vh2 = np.linalg.svd(rest.reshape(-1, d3), full_matrices=False)
    a1 = (u2 * s2).reshape(rest.shape[0], d3, -1)
    a2 = vh2.reshape(-1, d3, 1)
    mps = RefMPS(qs3, [a0, a1, a2])
    assert np.allclose(mps.contract(), psi)
    seq = [ref_cv.F(0), ref_cv.CZ(0, 1, 0.6, **exact), ref_cv.X(1, 0.5), ref_cv.P(2, 0.3),
           ref_cv.CZ(2, 1, 0.4, dagger=True, **exact), ref_cv.Z(0, 1.1), ref_cv.F(2, dagger=True)]
    for g in seq:
        g.apply(mps, rng=None)
    arrays["cv_seq_out"] = mps.contract()
    cases.append({"kind": "sequence", "key": "cv_seq_out",
                  "gates": ["F(0)", "CZ(0,1,0.6)", "X(1,0.5)", "P(2,0.3)", "CZ(2,1,0.4)^dag", "Z(0,1.1)", "F(2)^dag"]})
    save("cv_operators.npz", cases=json.dumps(cases), **arrays)


# (7b) cv_simulator: state preparation, homodyne measurements with forced results, Insert, the Simulator loop ----
def gen_cv_extra():
    from simulators.cv_simulator.simulator import Simulator as RefCVSimulator
    from simulators.cv_simulator.states import State as RefCVState

    arrays, cases = {}, []
    qs = np.linspace(-9.0, 9.0, 64)
    arrays["qs64"] = qs
    for name, eps in [("VACUUM", None), ("GKP_ZERO", 0.2), ("GKP_ONE", 0.2), ("GKP_PLUS", 0.15), ("GKP_T", 0.1),
                      ("GKP_H", 0.3), ("GKP_MINUS", 0.25), ("GKP_TDG", 0.2), ("QUNAUGHT", 0.2)]:
        key = f"state_{name}"
        arrays[key] = np.asarray(RefCVState[name].eval(qs, eps), dtype=np.complex128)
        cases.append({"kind": "state", "key": key, "name": name, "eps": eps})

    exact = {"rel_err": 0.0, "abs_err": 0.0}
    rng = np.random.default_rng(17)
    d = 16
    qs16 = np.linspace(-6.0, 6.0, d)
    arrays["qs16"] = qs16

    def exact_mps(psi):
        """Reference MPS holding ``psi`` exactly (successive SVDs, nothing truncated)."""
        n_modes = psi.ndim
        sites, rest, chi = [], psi.reshape(1, -1), 1
        for _ in range(n_modes - 1):
            u, s, vh = np.linalg.svd(rest.reshape(chi * d, -1), full_matrices=False)
            sites.append(u.reshape(chi, d, -1))
            chi = u.shape[1]
            rest = s[:, None] * vh
        sites.append(rest.reshape(chi, d, 1))
        mps = RefMPS(qs16, sites)
        assert np.allclose(mps.contract(), psi)
        return mps

    for n_modes in (2, 3):
        psi = rng.standard_normal((d,) * n_modes) + 1j * rng.standard_normal((d,) * n_modes)
        psi /= np.sqrt(np.sum(np.abs(psi) ** 2) * ((qs16[-1] - qs16[0]) / (d - 1)) ** n_modes)
        arrays[f"meas_in_{n_modes}"] = psi
        for label, make in [("Mq", lambda i, r: ref_cv.Mq(i, r)), ("Mp", lambda i, r: ref_cv.Mp(i, r)),
                            ("Hom0.7", lambda i, r: ref_cv.Homodyne(i, 0.7, r)),
                            ("Hompi", lambda i, r: ref_cv.Homodyne(i, np.pi, r))]:
            for index in range(n_modes):
                for forced in (-1.3, 0.45):
                    mps = exact_mps(psi)
                    res = make(index, forced).apply(mps, rng=None)
                    key = f"meas_{label}_{n_modes}_{index}_{forced}"
                    arrays[key] = np.asarray(mps.contract(), dtype=np.complex128)
                    cases.append({"kind": "measure", "key": key, "gate": label, "n_modes": n_modes, "index": index,
                                  "forced": forced, "result": float(res.result), "probability": float(res.probability)})

    # Insert chain + gates through the reference's Simulator (truncation off), vacuum / GKP inputs
    gates = [ref_cv.Insert(0, RefCVState.VACUUM), ref_cv.Insert(1, RefCVState.GKP_PLUS, gkp_epsilon=0.3),
             ref_cv.Insert(1, RefCVState.GKP_ZERO, gkp_epsilon=0.3), ref_cv.X(0, 0.4), ref_cv.CZ(0, 1, 0.5, **exact),
             ref_cv.F(2), ref_cv.BS(1, 2, 0.6, **exact), ref_cv.P(1, 0.2), ref_cv.CX(1, 0, 0.5, **exact),
             ref_cv.SWAP(0, 1, **exact), ref_cv.Homodyne(2, 0.4, 0.8), ref_cv.D(0, [0.3, -0.2]), ref_cv.Mp(0, -0.5)]
    sim = RefCVSimulator(gates, rng_seed=3, svd_options={"rel_err": 0.0})
    out = sim.run(RefMPS(qs16, []))
    arrays["sim_out"] = np.asarray(out.contract(), dtype=np.complex128)
    arrays["sim_results"] = np.array([[r.result, r.probability] for r in sim.results])
    cases.append({"kind": "simulator", "key": "sim_out"})
    save("cv_extra.npz", cases=json.dumps(cases), **arrays)


def gen_cv_mps():
    """Matrix-product states WITH truncation through the reference's gates: site shapes after every gate, checkpoints
    of the contracted state, norms and measurement records.  Grid and caps are chosen so that the
    reference stays on its exact-SVD branch (max_bond_dim * 10 >= min(matrix shape), mps.py:78)."""
    from simulators.cv_simulator.states import State as RefCVState
    from simulators.cv_simulator.mps import tensor_svd as ref_tensor_svd

    arrays, cases = {}, []
    d = 20
    qs = np.linspace(-6.5, 6.5, d)
    arrays["qs"] = qs
    for label, options in [("rel1e-6", {"rel_err": 1e-6}), ("cap5", {"max_bond_dim": 5}),
                           ("abs1e-3", {"abs_err": 1e-3, "rel_err": 0.0})]:
        mps = RefMPS(qs, [])
        rng = np.random.default_rng(5)
        shapes, norms, results = [], [], []
        for position, gate in enumerate(cv_mps_program(ref_cv, RefCVState, options)):
            out = gate.apply(mps, rng=rng)
            if out is not None and hasattr(out, "probability"):
                results.append([position, out.result, out.probability])
            shapes.append([list(t.shape) for t in mps.tensors])
            norms.append(float(np.real(mps.norm())))
            if position in (6, 9, 14, 18):
                arrays[f"{label}_state_{position}"] = np.asarray(mps.contract(), dtype=np.complex128)
        arrays[f"{label}_norms"] = np.array(norms)
        arrays[f"{label}_results"] = np.array(results)
        arrays[f"{label}_marginal"] = np.real(np.diag(mps.partial_density_mps(1)))
        arrays[f"{label}_rho0"] = np.asarray(mps.partial_density_mps(0), dtype=np.complex128)
        cases.append({"label": label, "options": options, "shapes": shapes})

    # tensor_svd itself on random tensors (exact branch), all three truncation modes
    rng = np.random.default_rng(23)
    for idx, (shape, left, right, options) in enumerate([
            ((3, 8, 8, 2), [0, 1], [2, 3], {"rel_err": 1e-3}),
            ((4, 6, 6, 5), [0, 2], [1, 3], {"max_bond_dim": 7}),
            ((2, 9, 9, 3), [0, 1], [2, 3], {"abs_err": 0.8, "rel_err": 0.0}),
            ((5, 7, 7, 1), [0, 1], [2, 3], {})]):
        t = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        t *= np.exp(-0.4 * np.arange(shape[1]))[None, :, None, None]        # decaying spectrum so truncation bites
        m1, m2 = ref_tensor_svd(t, left, right, **options)
        arrays[f"svd_in_{idx}"] = t
        arrays[f"svd_product_{idx}"] = np.tensordot(m1, m2, axes=1)
        cases.append({"kind": "tensor_svd", "index": idx, "left": left, "right": right, "options": options,
                      "rank": int(m1.shape[-1])})
    save("cv_mps.npz", cases=json.dumps(cases), **arrays)


def import_reference_gkp():
    """``simulators.gkp_simulator`` needs Python >= 3.12 for two PEP 695 statements (``type Syndrome = ...``,
    ``type GKPEC = MBI``, gates.py:12,211); this container runs 3.10.  The loader below feeds the interpreter the file
    with those two statements spelled as plain assignments -- the same meaning -- in memory only."""
    import importlib.abc
    import importlib.machinery
    import importlib.util
    import re

    class Loader(importlib.machinery.SourceFileLoader):
        def get_data(self, path):
            data = super().get_data(path)
            if str(path).endswith(".py"):
                data = re.sub(rb"(?m)^type (\w+) = ", rb"\1 = ", data)
            return data

    class Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if not name.startswith("simulators.gkp_simulator"):
                return None
            rel = Path(*name.split("."))
            for candidate, is_pkg in ((REFERENCE / rel / "__init__.py", True), (REFERENCE / rel.with_suffix(".py"), False)):
                if candidate.exists():
                    return importlib.util.spec_from_file_location(
                        name, candidate, loader=Loader(name, str(candidate)),
                        submodule_search_locations=[str(candidate.parent)] if is_pkg else None)
            return None

    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, Finder())
    from simulators.gkp_simulator import gates as g, insert_bell as b, simulator as s, transpiler as t, utils as u
    return g, b, s, t, u


def gen_gkp():
    """Measurement-based GKP layer: utilities, layering, frame commutation, gadgets with forced
    outcomes, whole seeded simulations and logical read-out, all from the reference run through the loader above."""
    g, b, sim_mod, t, u = import_reference_gkp()
    from simulators.cv_simulator.simulator import Simulator as RefCVSimulator

    arrays, cases = {}, {}
    cases["eps2db"] = [[e, float(u.eps2db(e))] for e in (0.05, 0.1, 0.3, 0.7)]
    cases["db2eps"] = [[x, float(u.db2eps(x))] for x in (6.0, 10.0, 14.5)]
    cases["format_result"] = [[v, u.format_result(v)] for v in (0.0, 1.3, -2.71, 4.0)]
    cases["cv2dv"] = [[v, bool(u.cv2dv_information(v))] for v in (0.1, 1.7, -1.9, 3.6, 5.2)]
    arrays["syndrome_matrix"] = np.asarray(u.syndrome_matrix([(1, 0), (0, 1), (1, 1)]), dtype=np.complex128)

    programs = gkp_programs(ref_gates)
    cases["layering"] = {}
    for name, gates in programs.items():
        circ = t.MBGKPCircuit.transpile(gates)
        filled = t.MBGKPCircuit.transpile(gates)
        filled.fill()
        cases["layering"][name] = {"text": circ.to_string(), "depth": circ.depth(), "count": circ.count(),
                                   "filled": filled.to_string(), "filled_count": filled.count()}

    frames = [[(0, 0), (1, 0)], [(1, 1), (0, 1)], [(1, 0), (1, 1)]]
    table = []
    for label, make in [("I", lambda: ref_gates.I(0)), ("T", lambda: ref_gates.T(0)), ("Tdg", lambda: ref_gates.Tdg(1)),
                        ("H", lambda: ref_gates.H(1)), ("P", lambda: ref_gates.P(0)), ("Pdg", lambda: ref_gates.Pdg(1)),
                        ("CZ", lambda: ref_gates.CZ(0, 1)), ("SWAP", lambda: ref_gates.SWAP(1, 0))]:
        for frame in frames:
            out_frame, out_gate = sim_mod.commute(make(), frame)
            table.append({"gate": label, "frame": frame, "out": [list(p) for p in out_frame], "applied": repr(out_gate)})
    cases["commute"] = table

    syn = []
    rng = np.random.default_rng(31)
    for label, make in [("MBI", lambda: g.MBI(0)), ("MBF", lambda: g.MBF(0)), ("MBFdg", lambda: g.MBF(0, dagger=True)),
                        ("MBP", lambda: g.MBP(1)), ("MBPdg", lambda: g.MBP(1, dagger=True)), ("MBT", lambda: g.MBT(0)),
                        ("MBTdg", lambda: g.MBT(0, dagger=True)), ("MBCZ", lambda: g.MBCZ(0, 1)),
                        ("MBSWAP", lambda: g.MBSWAP(2, 1))]:
        gadget = make()
        for _ in range(4):
            results = list(rng.normal(0, 2.5, size=2 * len(gadget.indices)))
            out, idx = gadget.compute_syndrome(results)
            syn.append({"gadget": label, "results": results, "syndromes": [list(x) for x in out], "indices": list(idx),
                        "compiled": [repr(c) for c in gadget.compile()], "angles": [float(a) for a in gadget.angles()]})
    cases["syndromes"] = syn

    # --- numerics on a small grid ---------------------------------------------------------------------------
    d, eps = 60, 0.4
    qs = np.linspace(-8.5, 8.5, d)
    arrays["qs"] = qs
    options = {"rel_err": 1e-9}
    for name in ("PLUS", "T", "Tdg"):
        arrays[f"bell_{name}"] = np.asarray(b.GKPBellState[name].eval(qs, eps).contract(), dtype=np.complex128)

    # InsertBell in the middle / at the ends of a chain
    from simulators.cv_simulator.states import State as RefCVState
    qs_small = np.linspace(-6.0, 6.0, 14)          # four modes are contracted below: keep the fixture small
    arrays["qs_small"] = qs_small
    chain = RefMPS(qs_small, [RefCVState.GKP_PLUS.eval(qs_small, eps), RefCVState.GKP_ZERO.eval(qs_small, eps)])
    ref_cv.CZ(0, 1, 1.0, **options).apply(chain)
    b.InsertBell(1, b.GKPBellState.T, gkp_epsilon=eps, **options).apply(chain, rng=None)
    arrays["insert_bell_mid"] = np.asarray(chain.contract(), dtype=np.complex128)
    cases["insert_bell_mid_shapes"] = [list(x.shape) for x in chain.tensors]

    # gadgets with forced outcomes on code-word inputs
    forced = []
    for label, make, inputs in [
            ("MBF", lambda r: g.MBF(0, eps, results=r, **options), ["GKP_ZERO"]),
            ("MBP", lambda r: g.MBP(1, eps, results=r, **options), ["GKP_PLUS", "GKP_PLUS"]),
            ("MBT", lambda r: g.MBT(0, eps, results=r, **options), ["GKP_PLUS", "GKP_ZERO"]),
            ("MBTdg", lambda r: g.MBT(0, eps, results=r, dagger=True, **options), ["GKP_PLUS"]),
            ("MBCZ", lambda r: g.MBCZ(0, 1, eps, results=r, **options), ["GKP_PLUS", "GKP_PLUS"]),
            ("MBSWAP", lambda r: g.MBSWAP(1, 0, eps, results=r, **options), ["GKP_ZERO", "GKP_PLUS"])]:
        n_meas = 2 if label in ("MBF", "MBP", "MBT", "MBTdg") else 4
        results = tuple(float(x) for x in rng.normal(0, 0.6, size=n_meas))
        gadget = make(results)
        mps = RefMPS(qs, [RefCVState[s].eval(qs, eps) for s in inputs])
        runner = RefCVSimulator(gadget.compile(), rng_seed=1)
        out = runner.run(mps)
        key = f"gadget_{label}"
        arrays[key] = np.asarray(out.contract(), dtype=np.complex128)
        forced.append({"gadget": label, "inputs": inputs, "results": list(results), "key": key,
                       "measured": [[r.result, r.probability] for r in runner.results],
                       "shapes": [list(x.shape) for x in out.tensors]})
    cases["forced_gadgets"] = forced

    # whole seeded simulations: register, frame, logical density matrix
    runs = []
    for name, inputs, seed in [("h_cz_p", ["ZERO", "PLUS"], 3), ("t_branch", ["PLUS", "ZERO"], 8),
                               ("swap_mix", ["ONE", "PLUS"], 5)]:
        circ = t.MBGKPCircuit.transpile(gkp_programs(ref_gates)[name])
        simulator = sim_mod.Simulator(circ, eps, rng_seed=seed, svd_options=options)
        init = t.parse_to_mps([RefState[s] for s in inputs], eps, qs)
        out, frame = simulator.run(init)
        arrays[f"run_{name}_state"] = np.asarray(out.contract(), dtype=np.complex128)
        arrays[f"run_{name}_rho"] = np.asarray(u.full_logical_density_mps(out), dtype=np.complex128)
        arrays[f"run_{name}_rho_normalised"] = np.asarray(u.full_logical_density_mps(out, normalised=True),
                                                          dtype=np.complex128)
        runs.append({"name": name, "inputs": inputs, "seed": seed, "frame": [list(p) for p in frame],
                     "shapes": [list(x.shape) for x in out.tensors]})
    alt = sim_mod.SimulatorAlt(t.MBGKPCircuit.transpile(gkp_programs(ref_gates)["h_cz_p"]), eps, rng_seed=4,
                               svd_options=options)
    out, frame = alt.run(t.parse_to_mps([RefState.ZERO, RefState.PLUS], eps, qs))
    arrays["run_alt_state"] = np.asarray(out.contract(), dtype=np.complex128)
    cases["alt_frame"] = [list(p) for p in frame]
    cases["runs"] = runs
    cases["eps"] = eps
    cases["options"] = options

    # read-out operators on product code words
    for n_modes, names in [(1, ["GKP_T"]), (2, ["GKP_H", "GKP_MINUS"])]:
        mps = RefMPS(qs, [RefCVState[s].eval(qs, eps) for s in names])
        arrays[f"rho_product_{n_modes}"] = np.asarray(u.full_logical_density_mps(mps), dtype=np.complex128)
    save("gkp.npz", cases=json.dumps(cases), **arrays)


if __name__ == "__main__":
    import logging
    logging.getLogger("simulators").setLevel(logging.ERROR)
    if "--cv-extra-only" in sys.argv:
        gen_cv_extra()
        sys.exit(0)
    if "--cv-mps-only" in sys.argv:
        gen_cv_mps()
        sys.exit(0)
    if "--gkp-only" in sys.argv:
        gen_gkp()
        sys.exit(0)
    if "--readout-only" in sys.argv:
        gen_readout()
        sys.exit(0)
    gen_single_gates()
    gen_expand_gate()
    gen_clifford()
    gen_random_circuits()
    gen_measure_insert()
    gen_readout()
    gen_grover()
    gen_cv()
    gen_cv_extra()
    gen_cv_mps()
    gen_gkp()
