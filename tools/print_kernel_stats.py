"""Print a rocprofv3 kernel_stats.csv (found under the given directory) as name / calls / average ns / percent."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-64s %6s %12s %6s" % (r["Name"][:64], r["Calls"], r["AverageNs"], r["Percentage"]))
