"""Print every (kernel, counter, value) row of the rocprofv3 counter-collection CSVs under the given directories.
Optional first argument `--kernel SUBSTR` keeps only kernels whose name contains SUBSTR (default: k_transport or k_chase)."""
import csv, glob, sys
args = sys.argv[1:]
keep = ('k_transport', 'k_chase')
if args and args[0] == '--kernel':
    keep = (args[1],); args = args[2:]
for d in args:
    for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
        for row in csv.DictReader(open(f)):
            if any(s in row['Kernel_Name'] for s in keep):
                dur = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6 if 'End_Timestamp' in row else float('nan')
                print('%-24s %-28s %18.0f   (%.2f ms)  grid %s' % (d.split('/')[-1], row['Counter_Name'], float(row['Counter_Value']), dur, row.get('Grid_Size', '?')))
