"""Host stand-ins used to DRIVE the product without a webui -- not the checker, and no arithmetic of the path.

  stub_host.py    a minimal A1111 (stable-diffusion-webui) module tree: lets the plugin (and, in the build container, the upstream
                  reference) be imported and driven from tests, bench.py and __graft_entry__.smoke()
  ldm_decoder.py  the `nn.Module` definition of the SD / SDXL KL-f8 auto-encoder (ldm.modules.diffusionmodules.model) with seeded
                  random weights: the object whose `forward` the Tiled-VAE hook replaces, and the holder of the weights

They lived under oracle/ until round 4; they were moved so that "the timed path of bench.py never imports oracle/" can be
checked with grep: oracle/ now holds only the CPU restatement of the reference's arithmetic (the checker).
"""
