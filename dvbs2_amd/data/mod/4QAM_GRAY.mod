# 4QAM avec codage de Gray
1	1
-1	1
1	-1
-1	-1
