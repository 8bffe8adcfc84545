"""Print per-kernel register / LDS / spill figures from the metadata of a hipcc -S listing."""
import re
import sys

text = open(sys.argv[1] if len(sys.argv) > 1 else "t5vqvae_amd/csrc/t5vq_kernels.s").read()
meta = text[text.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = g("name")
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    print(f"{name[:60]:60s} vgpr={g('vgpr_count'):>4s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} "
          f"lds={g('group_segment_fixed_size'):>6s} scratch={g('private_segment_fixed_size'):>4s} "
          f"spill={g('vgpr_spill_count')}")
