# A/B of whole-library builds on one box: profiles/tmp_libs/<name>.so are copied over
# root_digger_amd/lib/librdamd.so in turn.   LIBS="base skip" CMD="bash profiles/fd_ab.sh" bash profiles/lib_ab.sh
cp root_digger_amd/lib/librdamd.so /tmp/librdamd_keep.so
for rep in 1 2; do for l in $LIBS; do cp profiles/tmp_libs/$l.so root_digger_amd/lib/librdamd.so; echo "== $l"; eval "$CMD"; done; done
cp /tmp/librdamd_keep.so root_digger_amd/lib/librdamd.so
