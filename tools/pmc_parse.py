import csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'k_transport' in row['Kernel_Name']:
                print('%-28s %18.0f   (%.2f ms)' % (row['Counter_Name'], float(row['Counter_Value']), (int(row['End_Timestamp'])-int(row['Start_Timestamp']))/1e6))
