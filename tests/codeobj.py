"""Test helper: per-kernel register / scratch metadata of a HIP object or shared library.

The gfx950 code object sits in the `.hip_fatbin` section as a clang offload bundle; its AMDGPU
metadata note (llvm-readelf --notes) lists, per kernel, `.vgpr_count`, `.sgpr_count`,
`.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per lane)
and `.group_segment_fixed_size` (static LDS).  Needs only binutils' objcopy and the LLVM tools
that ship with ROCm -- no GPU."""
import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def tools_available():
    return (shutil.which("objcopy") is not None and os.path.exists(os.path.join(LLVM_BIN, "clang-offload-bundler"))
            and os.path.exists(os.path.join(LLVM_BIN, "llvm-readelf")))


def _demangle(names):
    filt = os.path.join(LLVM_BIN, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt")
    if not filt or not names:
        return {n: n for n in names}
    out = subprocess.run([filt] + list(names), check=True, capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def kernel_metadata(path):
    """{demangled kernel name: {field: int}} for every gfx950 kernel in `path` (.o or .so)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]  # a linked .so holds one bundle per TU
        out = {}
        for bi, st in enumerate(starts):
            end = starts[bi + 1] if bi + 1 < len(starts) else len(blob)
            one = os.path.join(tmp, f"bundle{bi}.bin")
            open(one, "wb").write(blob[st:end])
            co = os.path.join(tmp, f"dev{bi}.co")
            r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={one}",
                                f"--targets={TARGET}", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            for chunk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                chunk = ".agpr_count" + chunk
                name = re.search(r"\.name:\s+(\S+)", chunk)
                if not name:
                    continue
                rec = {}
                for f in FIELDS:
                    m = re.search(r"\.%s:\s+(\d+)" % f, chunk)
                    rec[f] = int(m.group(1)) if m else None
                out[name.group(1)] = rec
    dm = _demangle(list(out))
    return {dm[k]: v for k, v in out.items()}


if __name__ == "__main__":
    import sys

    for name, rec in sorted(kernel_metadata(sys.argv[1]).items()):
        flt = sys.argv[2] if len(sys.argv) > 2 else ""
        if flt in name:
            print(f"{name[:110]:110s} vgpr {rec['vgpr_count']:3d} agpr {rec['agpr_count']:3d} sgpr {rec['sgpr_count']:3d} "
                  f"vspill {rec['vgpr_spill_count']:3d} sspill {rec['sgpr_spill_count']:3d} scratch {rec['private_segment_fixed_size']:4d} "
                  f"lds {rec['group_segment_fixed_size']}")
