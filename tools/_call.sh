echo "# tools/layer_bench.py --iters 10: the 13 SingleConv shapes of UNet3D f_maps=32 launched in isolation through the C-ABI (decoder first convs on the GENERAL"
echo "# virtual-concat kernels here; the model runs their upsampled half on the sub-pixel kernels).  variants fwd/dgrad/wgrad: u3d_conv3d_variant"
echo "# (0 generic, 1 persistent, 2 persistent ragged, 3 split-K) / u3d_conv3d_wgrad_variant (0 generic staging, 1 / 2 constant-offset staging without / with ragged tiles, |4 tap pairs)"
for cfg in "1 80,170,170" "1 80,168,168" "1 64,128,128" "2 64,128,128"; do set -- $cfg; echo; echo "## batch $1 patch $2"; python tools/layer_bench.py --batch $1 --patch $2 --iters 10 2>/dev/null; done
