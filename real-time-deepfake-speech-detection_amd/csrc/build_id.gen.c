const char afx_build_id_str[] = "5d02bacc80a6";
