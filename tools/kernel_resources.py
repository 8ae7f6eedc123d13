"""Dev tool: per-kernel register / LDS / spill table from the gfx950 ISA metadata (hipcc -save-temps), i.e. how many
blocks of each kernel a CU can hold -- the input of the "rounds" analysis in DESIGN.md.   python tools/kernel_resources.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "retrieval-based-voice-conversion-webui_amd", "csrc")
KEEP = ("k_lm_", "k_rb_pair", "k_rb_full", "k_rb_stream", "k_frame_rms", "k_change_rms", "k_ups", "k_conv_mfma", "k_post", "k_scan", "k_coarse", "k_blend", "k_fr_", "k_sola", "k_f0_post",
        "k_rmvpe", "k_phase_scan", "k_sine")
with tempfile.TemporaryDirectory() as tmp:
    for src in ("nsf.hip", "rb_stream.hip", "ivf.hip", "front.hip", "glue.hip"):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-O3", "-std=c++17", "-c", os.path.join(CSRC, src), "-o",
                        os.path.join(tmp, src + ".o"), "-save-temps=obj"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.startswith(src.split(".")[0] + "-hip-amdgcn") and f.endswith(".s")]
        if not asm:
            continue
        s = open(os.path.join(tmp, asm[0])).read()
        print("== %s" % src)
        print("%-78s %5s %5s %6s %7s" % ("kernel (demangled prefix)", "vgpr", "agpr", "spill", "ldsB"))
        for b in s.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            if not any(k in name for k in KEEP) or "DF16b" in name:
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("rvcmi::", "").replace("void ", "").replace("(anonymous namespace)::", "")
            v = re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)
            lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", b).group(1)
            print("%-78s %5s %5s %6s %7s" % (dem[:78], v, b.split("\n")[0].strip(), sp, lds))
