import pandas as pd, glob, sys
f=(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')+glob.glob(sys.argv[1]+'/*kernel_trace.csv'))[0]
df=pd.read_csv(f)
df['name']=df['Kernel_Name'].str.replace(r'\(.*','',regex=True).str.replace('void ','').str.slice(0,28)
df=df.sort_values('Start_Timestamp').reset_index(drop=True)
t0=df['Start_Timestamp'].iloc[0]
mid=df[df['Start_Timestamp']>t0+ (df['End_Timestamp'].max()-t0)*0.9].head(int(sys.argv[2]) if len(sys.argv)>2 else 30)
b=mid['Start_Timestamp'].iloc[0]
for _,r in mid.iterrows():
    print("%-30s q%-3d start %8.1f end %8.1f dur %7.1f" % (r['name'], r['Queue_Id'], (r['Start_Timestamp']-b)/1e3, (r['End_Timestamp']-b)/1e3, (r['End_Timestamp']-r['Start_Timestamp'])/1e3))
