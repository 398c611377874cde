testFiles/multi.fa -f testFiles/multi.fa -i -o testFiles/tmp
embedded

+++ Path Summary Report +++
pos	header	telomeres	labels	gaps	type	granular	its	canonical	windows
1	contig_t2t	2	pq	0	t2t	PQ	0	199	4
2	contig_none	0	none	0	none		0	0	2
3	contig_incomplete	1	q	0	incomplete	Q	0	100	3

+++ Assembly Summary Report +++
Total paths:	3
Total gaps:	0
Scaffold N50:	3000
Contig N50:	3000
Total telomeres:	3
Total ITS blocks:	0
Total canonical matches:	299
Total windows analyzed:	9

+++ Telomere Statistics +++
Mean length:	600
Median length:	600
Min length:	600
Max length:	600

+++ Chromosome Telomere Counts+++
Two telomeres:	1
One telomere:	1
Zero telomeres:	1

+++ Chromosome Telomere/Gap Completeness+++
T2T:	1
Gapped T2T:	0
Misassembled:	0
Gapped misassembled:	0
Incomplete:	1
Gapped incomplete:	0
No telomeres:	1
Gapped no telomeres:	0
Discordant:	0
Gapped discordant:	0
