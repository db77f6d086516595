"""THE ONE COMMAND THAT TURNS THE IMAGE PATH'S PARITY FROM "UNPINNED" TO "PINNED" (last bullet).

It cannot run in the build container: the reference's image path needs kornia 0.7.0, torchvision 0.17.0 and e2cnn
(/root/reference/conda_env.yaml:44,230; setup.cfg:48-49), none of which exist there.  On ANY machine that has them:

    pip install kornia==0.7.0 torchvision==0.17.0 e2cnn omegaconf          # (torch 2.2.0 is what the reference pins them against)
    python tests/golden/regen_with_reference.py --reference /path/to/an/unmodified/equiadapt/checkout

What it does:
  1. REFUSES to run unless kornia 0.7.0 and torchvision 0.17.0 (and omegaconf) import with exactly those versions
     (``--allow-other-versions`` proceeds and stamps the versions it found into the provenance string; e2cnn absent: the
     network cases are skipped and say so).
  2. Imports the UNMODIFIED ``equiadapt`` package from ``--reference`` (sys.path, no stand-in modules of any kind).
  3. Regenerates every case of ``tests/golden/images_restatement.pt`` -- C8 and D4 canonicalize, invert (regular / scalar),
     the pre-canonicalization crop + resize, the D4 orbit, grayscale C4, mask rotation -- THROUGH THE REFERENCE'S OWN CLASSES
     (GroupEquivariantImageCanonicalization / OptimizedGroupEquivariantImageCanonicalization around a network that returns
     prescribed group activations; equiadapt.images.utils.rotate_masks), on the inputs stored in that fixture.
  4. Builds one ``ESCNNEquivariantNetwork`` of the reference (e2cnn), ``export()``s its layers (R2Conv -> nn.Conv2d,
     InnerBatchNorm -> nn.BatchNorm2d), runs it on a fixed batch, and runs this repository's ``ESCNNEquivariantNetwork`` with
     ``load_exported_dense`` on the same exported layers (CPU: its plain-module path; the HIP fast path is checked against
     that path by tests/test_gpu_parity.py).
  5. Prints max / rms differences of every case against ``oracle/`` (and against the restatement fixture itself), writes
     ``tests/golden/images_reference.pt`` (data only, provenance "reference (kornia x, torchvision y, e2cnn z)"), and exits
     non-zero if a difference exceeds the tolerance the parity tests use (2.8e-4 max / 2.4e-5 rms on unit-variance pixels,
     bit-exact masks, 1e-5 on the network activations).

Once ``images_reference.pt`` exists, ``tests/test_oracle_golden.py::test_oracle_matches_reference_generated_image_vectors``
(skipped while the file is absent) checks ``oracle/image_ops.py`` against it in every CPU run, and the header of
``oracle/image_ops.py`` may drop "PARITY UNPINNED".

``--dry-run`` prints the plan and the versions found without importing the reference (tests/test_abi_and_host.py runs it and
the refusal path).
"""
import argparse
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REQUIRED = {"kornia": "0.7.0", "torchvision": "0.17.0"}
PIX_MAX, PIX_RMS = 2.8e-4, 2.4e-5        # tests/test_gpu_parity.py: the white-noise pixel tolerance between two fp32 evaluations
ACT_TOL = 1e-5


class Refused(RuntimeError):
    pass


def check_versions(allow_other: bool) -> dict:
    """{package: version} of kornia / torchvision / omegaconf / e2cnn (None: absent).  Raises `Refused` per the rule above."""
    found = {}
    for name in ("kornia", "torchvision", "omegaconf", "e2cnn"):
        try:
            mod = importlib.import_module(name)
            found[name] = getattr(mod, "__version__", "unknown")
        except Exception:  # noqa: BLE001 -- ImportError, or a broken install: both mean "not usable"
            found[name] = None
    problems = []
    for name, want in REQUIRED.items():
        have = found[name]
        if have is None:
            problems.append(f"{name} is not installed (need {want})")
        elif have.split("+")[0] != want and not allow_other:
            problems.append(f"{name} {have} found, the reference pins {want} (--allow-other-versions to proceed and stamp the provenance)")
    if found["omegaconf"] is None:
        problems.append("omegaconf is not installed (the reference's constructors take a DictConfig)")
    if problems:
        raise Refused("regen_with_reference.py refuses to run: " + "; ".join(problems) +
                      ".  This script exists to produce REFERENCE-generated vectors; with stand-ins it would only restate the oracle.")
    return found


class PrescribedActivations(torch.nn.Module):
    """A 'canonicalization network' that answers with prescribed group activations (a margin of 10 on the wanted element): the
    reference's own argmax / one-hot / element-dictionary code then selects exactly the elements of the fixture."""

    def __init__(self, gidx: torch.Tensor, num_group: int, group_type: str, num_rotations: int):
        super().__init__()
        self.gidx, self.num_group = gidx, num_group
        self.group_type, self.num_rotations = group_type, num_rotations

    def forward(self, x):
        acts = torch.zeros(x.shape[0], self.num_group)
        acts[torch.arange(x.shape[0]), self.gidx[: x.shape[0]]] = 10.0
        return acts


def diff(name: str, got: torch.Tensor, want: torch.Tensor, out: list, exact: bool = False) -> bool:
    d = (got.double() - want.double()).abs()
    mx, rms = d.max().item(), d.pow(2).mean().sqrt().item()
    ok = (mx == 0.0) if exact else (mx <= PIX_MAX and rms <= PIX_RMS)
    out.append((name, mx, rms, ok))
    print(f"  {name:<44} max {mx:.3e}  rms {rms:.3e}  {'ok' if ok else 'OUTSIDE TOLERANCE'}")
    return ok


def run(reference: str, found: dict, out_path: str) -> int:
    sys.path.insert(0, reference)
    sys.path.insert(0, ROOT)
    import equiadapt                                                     # the UNMODIFIED reference package
    from equiadapt.images.canonicalization.discrete_group import (GroupEquivariantImageCanonicalization,
                                                                  OptimizedGroupEquivariantImageCanonicalization)
    from equiadapt.images.utils import rotate_masks
    from omegaconf import OmegaConf

    from oracle import image_ops as io

    assert os.path.realpath(os.path.dirname(equiadapt.__file__)).startswith(os.path.realpath(reference)), \
        f"`equiadapt` resolved to {equiadapt.__file__}, not to --reference {reference}"
    fix = torch.load(os.path.join(HERE, "images_restatement.pt"), weights_only=False)
    prov = "reference (" + ", ".join(f"{k} {v}" for k, v in found.items() if v) + ")"
    ref = {"provenance": prov, "x": fix["x"]}
    rows: list = []
    x = fix["x"]
    print(f"reference: {os.path.dirname(equiadapt.__file__)}\nprovenance: {prov}\ncase: |reference - oracle|")

    def canonicalizer(gidx, group_type, num_rotations, in_shape, crop=1.0, resize=None):
        G = num_rotations if group_type == "rotation" else 2 * num_rotations
        hp = OmegaConf.create({"beta": 1.0, "input_crop_ratio": crop, "resize_shape": resize or in_shape[-1]})
        can = GroupEquivariantImageCanonicalization(PrescribedActivations(gidx, G, group_type, num_rotations), hp, in_shape)
        return can.eval()

    for key, group_type, N in (("c8", "rotation", 8), ("d4", "roto-reflection", 4)):
        c = fix[key]
        can = canonicalizer(c["gidx"], group_type, N, (3, 32, 32))
        with torch.no_grad():
            canon = can.canonicalize(x)                                  # discrete_group.py:190-238
            el = can.canonicalization_info_dict["group_element"]
            inv_reg = can.invert_canonicalization(c["f"], induced_rep_type="regular")            # :240-259 -> images/utils.py:8-94
            inv_sca = can.invert_canonicalization(c["f"][:, :3], induced_rep_type="scalar")
        ref[key] = {"gidx": c["gidx"], "rotation": el["rotation"].clone(), "reflection": el.get("reflection", torch.zeros(0)).clone(),
                    "canon": canon, "f": c["f"], "invert_regular": inv_reg, "invert_scalar": inv_sca}
        ang = io.group_angles(N)
        ang = (torch.cat([ang, ang]) if group_type != "rotation" else ang)[c["gidx"]]
        refl = (c["gidx"] >= N).float() if group_type != "rotation" else None
        G = N if group_type == "rotation" else 2 * N
        diff(f"{key} canonicalize", canon, io.canonicalize_images(x, ang, refl, (3, 32, 32)), rows)
        diff(f"{key} invert regular", inv_reg, io.invert_action(c["f"], ang, refl, N, G, "regular"), rows)
        diff(f"{key} invert scalar", inv_sca, io.invert_action(c["f"][:, :3], ang, refl, N, G, "scalar"), rows)
        diff(f"{key} canonicalize vs restatement fixture", canon, c["canon"], rows)
    # pre-canonicalization transform (crop 0.8, resize 16): discrete_group.py:174-188 with :76-92
    can = canonicalizer(torch.zeros(4, dtype=torch.long), "rotation", 8, (3, 32, 32), crop=fix["pre"]["crop_ratio"], resize=fix["pre"]["resize"])
    with torch.no_grad():
        pre = can.transformations_before_canonicalization_network_forward(x)
    ref["pre"] = {"crop_ratio": fix["pre"]["crop_ratio"], "resize": fix["pre"]["resize"], "out": pre}
    diff("pre-canonicalization crop + resize", pre, io.pre_canonicalization_transform(x, (3, 32, 32), 0.8, 16), rows)
    # D4 orbit: OptimizedGroupEquivariantImageCanonicalization.group_augment, discrete_group.py:387-427

    class VectorNet(torch.nn.Module):
        out_vector_size = 8

        def forward(self, t):
            return torch.zeros(t.shape[0], 8)

    hp = OmegaConf.create({"beta": 1.0, "input_crop_ratio": 1.0, "resize_shape": 32, "group_type": "roto-reflection", "num_rotations": 4,
                           "artifact_err_wt": 0.0, "learn_ref_vec": False})
    opt = OptimizedGroupEquivariantImageCanonicalization(VectorNet(), hp, (3, 32, 32)).eval()
    opt.device = x.device
    with torch.no_grad():
        orbit = opt.group_augment(x[:2])
    ref["orbit_d4"] = orbit
    diff("D4 orbit (group_augment)", orbit, io.orbit_expand(x[:2], 4, "roto-reflection", 32), rows)
    # grayscale C4 (no pad / crop: discrete_group.py:61-72)
    can = canonicalizer(torch.tensor([0, 1, 2, 3]), "rotation", 4, (1, 32, 32))
    with torch.no_grad():
        gray = can.canonicalize(x[:, :1])
    ref["gray_c4"] = gray
    diff("grayscale C4 canonicalize", gray, io.canonicalize_images(x[:, :1], io.group_angles(4)[torch.tensor([0, 1, 2, 3])], None, (1, 32, 32)), rows)
    # masks: images/utils.py:120-140 (torchvision.transforms.functional.rotate, nearest)
    m = fix["masks"]["in"]
    ref["masks"] = {"in": m, "rot_m45": rotate_masks(m, -45.0), "rot_90": rotate_masks(m, 90.0)}
    diff("masks rotate -45", ref["masks"]["rot_m45"], io.rotate_masks(m, -45.0), rows, exact=True)
    diff("masks rotate +90", ref["masks"]["rot_90"], io.rotate_masks(m, 90.0), rows, exact=True)

    # e2cnn network through its exported dense layers
    if found.get("e2cnn"):
        import e2cnn
        from equiadapt.images.canonicalization_networks.escnn_networks import ESCNNEquivariantNetwork as RefNet

        import equiadapt_amd as ea

        torch.manual_seed(7)
        net = RefNet((3, 32, 32), 4, 5, "rotation", 8, 3).eval()        # 4 fields x 8 = 32 channels, k5, 3 layers
        xb = torch.randn(6, 3, 32, 32)
        with torch.no_grad():
            acts = net(xb)
        convs, norms = [], []
        for mod in net.eqv_network:
            if isinstance(mod, e2cnn.nn.R2Conv):
                convs.append(mod.export())
            elif isinstance(mod, e2cnn.nn.InnerBatchNorm):
                norms.append(mod.export())
        mine = ea.ESCNNEquivariantNetwork((3, 32, 32), 4, 5, "rotation", 8, 3).eval()
        mine.load_exported_dense(convs, norms)
        with torch.no_grad():
            got = mine(xb)
        d = (got - acts).abs().max().item()
        ok = d <= ACT_TOL * max(acts.abs().max().item(), 1.0)
        rows.append(("e2cnn network via load_exported_dense", d, float("nan"), ok))
        print(f"  {'e2cnn network via load_exported_dense':<44} max {d:.3e}  {'ok' if ok else 'OUTSIDE TOLERANCE'}  "
              f"(argmax equal: {bool(torch.equal(got.argmax(1), acts.argmax(1)))})")
        ref["escnn"] = {"x": xb, "activations": acts,
                        "convs": [{k: v.clone() for k, v in c.state_dict().items()} for c in convs],
                        "norms": [{k: v.clone() for k, v in n.state_dict().items()} for n in norms],
                        "ctor": {"in_shape": (3, 32, 32), "out_channels": 4, "kernel_size": 5, "group_type": "rotation", "num_rotations": 8,
                                 "num_layers": 3}}
    else:
        print("  e2cnn is not installed: the exported-network case is SKIPPED (image-action cases above are unaffected)")
    torch.save(ref, out_path)
    bad = [r for r in rows if not r[3]]
    print(f"wrote {out_path}: {len(rows)} comparisons, {len(bad)} outside tolerance")
    return 1 if bad else 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", help="root of an unmodified arnab39/equiadapt checkout (the directory that holds equiadapt/)")
    ap.add_argument("--out", default=os.path.join(HERE, "images_reference.pt"))
    ap.add_argument("--allow-other-versions", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="report the versions found and the plan; import nothing of the reference")
    a = ap.parse_args(argv)
    try:
        found = check_versions(a.allow_other_versions)
    except Refused as exc:
        print(str(exc), file=sys.stderr)
        if a.dry_run:
            print("dry run: would regenerate c8 / d4 canonicalize + invert (regular, scalar), pre-transform, D4 orbit, gray C4, masks, "
                  "and one exported e2cnn network -- AFTER the requirements above are met")
        return 2
    if a.dry_run:
        print("versions:", found)
        print("dry run: would regenerate c8 / d4 canonicalize + invert (regular, scalar), pre-transform, D4 orbit, gray C4, masks"
              + (", and one exported e2cnn network" if found.get("e2cnn") else " (e2cnn absent: network case skipped)"))
        return 0
    if not a.reference or not os.path.isdir(os.path.join(a.reference, "equiadapt")):
        print("--reference must point at a checkout that contains equiadapt/", file=sys.stderr)
        return 2
    return run(os.path.abspath(a.reference), found, a.out)


if __name__ == "__main__":
    sys.exit(main())
