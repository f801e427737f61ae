import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=max(int(r['Calls']) for r in rows if 'sdf_train_bwd' in r['Name'])
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps
launches=sum(int(r['Calls']) for r in rows)/steps
print(sys.argv[1], 'steps',steps,'kernel ms/step %.2f'%tot,'launches/step %.0f'%launches)
