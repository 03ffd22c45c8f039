import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); c=db.cursor()
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for r in c.execute(q): print("%-60s %5d avg %9.1f min %9.1f max %9.1f"%(r[0][:60],r[1],r[2],r[3],r[4]))
