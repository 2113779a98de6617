#4,10,50,50,-1
@0	tA
@1	tB
@2	tC
@3	tD
@4	tE
cid	no.tids	first.tid	other.tids	segment.length
0	1	0		98,
1	1	1		49,
2	1	2		98,
3	1	3		71,
4	1	4		98,
5	2	0	1,	142,
6	2	0	2,	71,
7	2	2	4,	71,
8	3	2	4,4,	71,
