for r in 1 2; do for lib in product clumped clumped_builtin; do
if [ $lib = product ]; then unset OTHELLO_MI355X_LIB; else export OTHELLO_MI355X_LIB=build/$lib/libothello_mi355x.so; fi
for n in 1 64 256; do python tools/netbench.py --n $n --nets 10x128x8:f16x3 2>&1 | grep -v amdgpu | sed "s|^|[$lib r$r] |"; done
done; done
