for v in abl1 abl2; do
  UQ_LIB_PATH=$PWD/uq_amd/_variants/libuqhip_$v.so UQ_MSDBENCH_ONE=DNA bash tools/prof_cmd.sh r04b/msd_$v tools/msdbench.py 50000000 auto > /dev/null 2>&1
  echo $v; grep "msd_finish" gpurun_out/r04b/msd_${v}_kernel_stats.txt
done
