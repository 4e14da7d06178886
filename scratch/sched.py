import re,sys
s=open(sys.argv[1]).read()
name=sys.argv[2]
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
f=s[i:j]
blocks=re.split(r'\n(?=\.LBB\d+_\d+:)',f)
for b in blocks:
    if 'v_mfma' in b:
        seq=[]
        for l in b.split('\n'):
            t=l.strip()
            if not t or t[0] in ';.' : continue
            op=t.split()[0]
            if op=='s_waitcnt': op=t.replace('\t',' ').replace('s_waitcnt ','W:')
            op=op.replace('v_mfma_f32_32x32x16_bf16','MFMA').replace('_e32','').replace('_e64','')
            seq.append(op)
        out=[];prev=None;n=0
        for o in seq:
            if o==prev:n+=1
            else:
                if prev: out.append(f"{prev}x{n}" if n>1 else prev)
                prev=o;n=1
        out.append(f"{prev}x{n}")
        print(b.split('\n')[0][:40], len(seq), ' | '.join(out)[:int(sys.argv[3]) if len(sys.argv)>3 else 2500]); print()
