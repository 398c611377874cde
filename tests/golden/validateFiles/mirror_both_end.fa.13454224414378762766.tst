testFiles/mirror_both_end.fa -f testFiles/mirror_both_end.fa -r -o testFiles/tmp
embedded

+++ Path Summary Report +++
pos	header	telomeres	labels	gaps	type	granular	its	canonical	windows
1	chr_mirror_both_end	2	pq	0	discordant	P*Q	0	199	4

+++ Assembly Summary Report +++
Total paths:	1
Total gaps:	0
Scaffold N50:	3200
Contig N50:	3200
Total telomeres:	2
Total ITS blocks:	0
Total canonical matches:	199
Total windows analyzed:	4

+++ Telomere Statistics +++
Mean length:	600
Median length:	600
Min length:	600
Max length:	600

+++ Chromosome Telomere Counts+++
Two telomeres:	0
One telomere:	0
Zero telomeres:	0

+++ Chromosome Telomere/Gap Completeness+++
T2T:	0
Gapped T2T:	0
Misassembled:	0
Gapped misassembled:	0
Incomplete:	0
Gapped incomplete:	0
No telomeres:	0
Gapped no telomeres:	0
Discordant:	1
Gapped discordant:	0
