import re,collections,sys
s=open('/root/repo/scratch/gemm.s').read()
name=sys.argv[1]
i=s.index(name+':'); j=s.index('.Lfunc_end',i)
f=s[i:j].split('\n')
blocks=[];cur=['entry',[]];blocks.append(cur)
for l in f:
    if re.match(r'^\.LBB\d+_\d+:',l): cur=[l,[]];blocks.append(cur)
    elif l.startswith('\t') and not l.strip().startswith(('.',';')): cur[1].append(l.strip().split()[0])
print(len(blocks),'blocks')
for b in blocks:
    c=collections.Counter(b[1])
    if len(b[1])>40:
        print(b[0][:60],len(b[1]),{k:v for k,v in c.most_common(16)})
