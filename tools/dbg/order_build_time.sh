# duration of k_order_count / _scan / _place under rocprofv3 --kernel-trace --stats (eighth-frame strips and the 4K frame)
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ob -o t -- python -c "import os,sys; os.chdir(os.environ[\"GRAFT_REPO_ROOT\"]); exec(open(\"tools/dbg/order_strip_ab.py\").read())" > /tmp/ob.log 2>&1
tail -1 /tmp/ob.log
find /tmp/ob -name "*kernel_stats.csv" > /tmp/ob.files
while read f; do grep -h "k_order_\|k_clouds<" "$f" < /dev/null | cut -c1-220; done < /tmp/ob.files
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d /tmp/ob2 -o t -- python $GRAFT_REPO_ROOT/tools/dbg/order_ab.py > /tmp/ob2.log 2>&1
tail -1 /tmp/ob2.log
find /tmp/ob2 -name "*kernel_stats.csv" > /tmp/ob2.files
while read f; do grep -h "k_order_" "$f" < /dev/null | cut -c1-220; done < /tmp/ob2.files
