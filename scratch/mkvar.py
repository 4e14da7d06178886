import re,os,shutil,subprocess,sys
sys.path.insert(0,'/root/repo')
from facl_amd import build
src='/root/repo/facl_amd/csrc'
def variant(tag, edits, fname='gemm.hip'):
    d=f'/root/repo/facl_amd/csrc_exp_{tag}'
    shutil.rmtree(d,ignore_errors=True); shutil.copytree(src,d)
    p=d+'/'+fname; s=open(p).read()
    for a,b in edits:
        assert a in s,(tag,a); s=s.replace(a,b)
    open(p,'w').write(s)
    build.build_variant(f'/root/repo/facl_amd/libfacl_hip_{tag}.so', d)
    shutil.copy(p,f'/root/repo/scratch/{tag}_{fname}')
    shutil.rmtree(d)
    print('built',tag)
