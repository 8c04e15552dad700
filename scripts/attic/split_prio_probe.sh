# split kernel with producers above consumers (experiment builds ab_libs/lib_sp{1,2,3}.so vs the shipped library)
for n in 100000 50000; do for v in base sp1 sp2 sp3; do
  if [ $v = base ]; then unset SCRG_LIB; else export SCRG_LIB=$PWD/ab_libs/lib_$v.so; fi
  echo "== $v $n"; python scripts/split_time.py $n 2>/dev/null | head -2
done; done
